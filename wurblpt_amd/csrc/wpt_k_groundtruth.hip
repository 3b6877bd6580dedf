/*
 * wpt_k_groundtruth.hip -- getGroundTruth (wurblpt.hpp:626-761): one ray through the centre of every
 * pixel, no randomness, and the geometry / material / flow buffers of its first hit.
 *
 * One lane = one pixel, one ray: the walk is the reference's BVH::hit in the stackless form the path
 * tracing kernels use, the hit record and the material tangent space are theirs too (wpt_device.h),
 * the ray comes from blockNew (wpt_blocks.h) with the pixel jitter and the lens sampling switched off.
 * The picture is taken at t0; the flow arrays compare the hit point's place at tPrev / tNext (it moves
 * with an animated instance) as seen by the camera at those times.
 */
#define WPT_SPHERE_HIT_INLINE /* see wpt_device.h: a real call would pin the kernel arguments to scratch memory */
#include "wpt_pathtrace.inc.h"

namespace wptk {

struct InverseTransformation {
    float rotation[4];
    f3 translation;
};

/* inverse(Transformation) (transformation.hpp:157-163); applied as rotation * p + translation (wurblpt.hpp:679) */
WPT_D InverseTransformation inverseOf(const wpt_camera& cam)
{
    InverseTransformation inv;
    inv.rotation[0] = -cam.rotation[0];
    inv.rotation[1] = -cam.rotation[1];
    inv.rotation[2] = -cam.rotation[2];
    inv.rotation[3] = cam.rotation[3];
    const f3 invT = neg(ld3(cam.translation));
    const f3 invS = mk3(1.0f / cam.scaling[0], 1.0f / cam.scaling[1], 1.0f / cam.scaling[2]);
    inv.translation = quatRotate(inv.rotation, mul(invT, invS));
    return inv;
}

/* Camera::cameraSpaceToImageSpace (camera.hpp:194-217) */
WPT_D f2 cameraSpaceToImageSpace(const wpt_camera& cam, f3 p)
{
    const float P00 = 2.0f / (cam.r - cam.l);
    const float P11 = 2.0f / (cam.t - cam.b);
    const float P03 = (cam.r + cam.l) / (cam.r - cam.l);
    const float P13 = (cam.t + cam.b) / (cam.t - cam.b);
    const float px = P00 * p.x + P03 * p.z;
    const float py = P11 * p.y + P13 * p.z;
    const float pw = -p.z;
    f2 ic;
    ic.x = 0.5f * (px / pw) + 0.5f;
    ic.y = 0.5f * (py / pw) + 0.5f;
    wptlens::distort(cam, ic.x, ic.y);
    return ic;
}

WPT_D void store3(void* array, uint32_t pixel, f3 v)
{
    if (array) {
        float* o = static_cast<float*>(array) + 3 * (size_t)pixel;
        o[0] = v.x;
        o[1] = v.y;
        o[2] = v.z;
    }
}
WPT_D void store2(void* array, uint32_t pixel, f2 v)
{
    if (array) {
        float* o = static_cast<float*>(array) + 2 * (size_t)pixel;
        o[0] = v.x;
        o[1] = v.y;
    }
}
WPT_D void store1(void* array, uint32_t pixel, float v)
{
    if (array)
        static_cast<float*>(array)[pixel] = v;
}

constexpr uint32_t GT_FEATURES = FEAT_TEXTURES | FEAT_LENS | FEAT_SPHERES | FEAT_ANIM;

__global__ void __launch_bounds__(256) wpt_ground_truth_kernel(const GroundTruthArgs args)
{
    const uint32_t pixel = blockIdx.x * blockDim.x + threadIdx.x;
    if (pixel >= args.width * args.height)
        return;
    const SceneView& sv = args.scene;

    /* the ray: pixel centre, pinhole (getRay(..., withRandomness = false), wurblpt.hpp:660-662) */
    FrameArgs fa;
    fa.cam = args.cam;
    fa.cam.lens_radius = 0.0f;
    fa.par = args.par;
    fa.par.randomize_ray_over_pixel = 0;
    fa.width = args.width;
    fa.height = args.height;
    fa.samplesSqrt = 1;
    fa.invWidth = 1.0f / (float)args.width;
    fa.invHeight = 1.0f / (float)args.height;
    fa.invSamplesSqrt = 1.0f;
    PathRegs ps;
    pathStateInit(ps, pixel, pixel % args.width, pixel / args.width);
    blockNew<GT_FEATURES>(fa, ps, sv); /* par.t0 == par.t1: no time draw, ps.time = t0 */

    /* BVH::hit (bvh.hpp:270-329): closest candidate, later candidates win ties */
    const RayAux aux = rayAux(ps.d);
    Candidate best;
    best.prim = NO_HIT;
    best.a = best.invDet = best.U = best.V = best.W = 0.0f;
    float amax = k_maxval;
    uint32_t node = 0;
    while (node < sv.nodeCount) {
        const float4 n0 = sv.nodes[2 * node], n1 = sv.nodes[2 * node + 1];
        const uint32_t skip = __float_as_uint(n1.z);
        const uint32_t prim = __float_as_uint(n1.w);
        const bool hit = boxTest(nodeLo(n0, n1), nodeHi(n0, n1), ps.o, aux.inv, args.par.min_hit_distance, amax);
        if (hit && prim < NODE_CHILD) {
            Candidate c;
            bool accepted;
            if (prim & PRIM_SPHERE) {
                c.invDet = c.U = c.V = c.W = 0.0f;
                accepted = sphereTest(sphereAt<GT_FEATURES>(sv, sv.spheres[prim & ~PRIM_SPHERE], ps.time), ps.o, ps.d, args.par.min_hit_distance, amax, c.a);
            } else {
                const float4 g0 = sv.triGeom[3 * (size_t)prim], g1 = sv.triGeom[3 * (size_t)prim + 1], g2 = sv.triGeom[3 * (size_t)prim + 2];
                f3 v0 = mk3(g0.x, g0.y, g0.z), v1 = mk3(g1.x, g1.y, g1.z), v2 = mk3(g2.x, g2.y, g2.z);
                if (__float_as_uint(g2.w) & WPT_TRI_ANIMATE) {
                    float animationM[16];
                    wptanim::toMat4(animationAt(sv, sv.instances[__float_as_uint(g0.w)].animation, ps.time), animationM);
                    v0 = animatePoint(animationM, v0);
                    v1 = animatePoint(animationM, v1);
                    v2 = animatePoint(animationM, v2);
                }
                accepted = triangleTest(v0, v1, v2, ps.o, aux, args.par.min_hit_distance, amax, c);
            }
            if (accepted) {
                c.prim = prim;
                best = c;
                amax = c.a;
            }
            node = skip; /* a leaf's subtree is the leaf itself */
        } else {
            node = hit ? (prim & NODE_INDEX_MASK) : skip;
        }
    }

    const f3 zero3 = mk3(0.0f, 0.0f, 0.0f);
    f3 wsPos = zero3, wsGNrm = zero3, wsGTan = zero3, wsMNrm = zero3, wsMTan = zero3;
    f3 csPos = zero3, csGNrm = zero3, csGTan = zero3, csMNrm = zero3, csMTan = zero3;
    float csDepth = 0.0f, csDist = 0.0f;
    f2 txCor;
    txCor.x = txCor.y = 0.0f;
    f3 wsOP = zero3, wsON = zero3, csOP = zero3, csON = zero3;
    f2 psOP = txCor, psON = txCor;
    int matInd = -1;
    if (best.prim != NO_HIT) {
        const Hit h = finishHit<GT_FEATURES>(sv, best, ps.o, ps.d, ps.time);
        wsPos = h.p;
        wsGNrm = h.n;
        wsGTan = h.t;
        /* the hitable's own material: a two-sided wrapper is not looked through (wurblpt.hpp:675) */
        const Frame ts = tangentSpaceAt<GT_FEATURES>(sv, sv.materials[h.material], h);
        wsMNrm = ts.n;
        wsMTan = ts.t;
        const InverseTransformation inv0 = inverseOf(args.cam);
        csPos = add(quatRotate(inv0.rotation, wsPos), inv0.translation);
        csGNrm = quatRotate(inv0.rotation, wsGNrm);
        csGTan = quatRotate(inv0.rotation, wsGTan);
        csMNrm = quatRotate(inv0.rotation, wsMNrm);
        csMTan = quatRotate(inv0.rotation, wsMTan);
        csDepth = -csPos.z;
        csDist = __builtin_sqrtf(dot(csPos, csPos));
        txCor = h.tc;
        f3 wsPosPrev = wsPos, wsPosNext = wsPos;
        int ai = -1;
        if (best.prim & PRIM_SPHERE) {
            ai = sv.spheres[best.prim & ~PRIM_SPHERE].animation;
        } else {
            const uint32_t inst = __float_as_uint(sv.triGeom[3 * (size_t)best.prim].w);
            if (__float_as_uint(sv.triGeom[3 * (size_t)best.prim + 2].w) & WPT_TRI_ANIMATE)
                ai = sv.instances[inst].animation;
        }
        if (ai >= 0) {
            /* wurblpt.hpp:695-699: back to where the instance keeps the point, then to where it is at tPrev / tNext */
            const wptanim::Trs T0 = animationAt(sv, ai, args.t0);
            wptanim::Trs inv;
            inv.q[0] = -T0.q[0];
            inv.q[1] = -T0.q[1];
            inv.q[2] = -T0.q[2];
            inv.q[3] = T0.q[3];
            for (int k = 0; k < 3; k++)
                inv.s[k] = 1.0f / T0.s[k];
            const f3 invT = quatRotate(inv.q, mul(neg(ld3(T0.t)), ld3(inv.s)));
            inv.t[0] = invT.x;
            inv.t[1] = invT.y;
            inv.t[2] = invT.z;
            const float pos[3] = { wsPos.x, wsPos.y, wsPos.z };
            float posOrig[3], moved[3];
            wptanim::applyTrs(inv, pos, posOrig);
            wptanim::applyTrs(animationAt(sv, ai, args.tPrev), posOrig, moved);
            wsPosPrev = mk3(moved[0], moved[1], moved[2]);
            wptanim::applyTrs(animationAt(sv, ai, args.tNext), posOrig, moved);
            wsPosNext = mk3(moved[0], moved[1], moved[2]);
        }
        wsOP = sub(wsPosPrev, wsPos);
        wsON = sub(wsPosNext, wsPos);
        const InverseTransformation invP = inverseOf(args.camPrev), invN = inverseOf(args.camNext);
        const f3 csPosPrev = add(quatRotate(invP.rotation, wsPosPrev), invP.translation);
        const f3 csPosNext = add(quatRotate(invN.rotation, wsPosNext), invN.translation);
        csOP = sub(csPosPrev, csPos);
        csON = sub(csPosNext, csPos);
        if (args.array[17] || args.array[18]) {
            const float fw = (float)args.width, fh = (float)args.height;
            f2 psPos;
            psPos.x = (((float)(pixel % args.width) + 0.5f) * (1.0f / fw)) * fw;
            psPos.y = (((float)(pixel / args.width) + 0.5f) * (1.0f / fh)) * fh;
            const f2 icPrev = cameraSpaceToImageSpace(args.cam, csPosPrev), icNext = cameraSpaceToImageSpace(args.cam, csPosNext);
            psOP.x = icPrev.x * fw - psPos.x;
            psOP.y = icPrev.y * fh - psPos.y;
            psON.x = icNext.x * fw - psPos.x;
            psON.y = icNext.y * fh - psPos.y;
        }
        matInd = (int)h.material;
    }
    store3(args.array[0], pixel, wsPos);
    store3(args.array[1], pixel, wsGNrm);
    store3(args.array[2], pixel, wsGTan);
    store3(args.array[3], pixel, wsMNrm);
    store3(args.array[4], pixel, wsMTan);
    store3(args.array[5], pixel, csPos);
    store3(args.array[6], pixel, csGNrm);
    store3(args.array[7], pixel, csGTan);
    store3(args.array[8], pixel, csMNrm);
    store3(args.array[9], pixel, csMTan);
    store1(args.array[10], pixel, csDepth);
    store1(args.array[11], pixel, csDist);
    store2(args.array[12], pixel, txCor);
    store3(args.array[13], pixel, wsOP);
    store3(args.array[14], pixel, wsON);
    store3(args.array[15], pixel, csOP);
    store3(args.array[16], pixel, csON);
    store2(args.array[17], pixel, psOP);
    store2(args.array[18], pixel, psON);
    if (args.array[19])
        static_cast<int32_t*>(args.array[19])[pixel] = matInd;
}

void launchGroundTruth(const GroundTruthArgs& args, hipStream_t stream)
{
    const uint32_t pixels = args.width * args.height;
    hipLaunchKernelGGL(wpt_ground_truth_kernel, dim3((pixels + 255) / 256), dim3(256), 0, stream, args);
}

}
