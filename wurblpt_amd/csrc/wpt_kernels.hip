/*
 * wpt_kernels.hip -- the gfx950 path-tracing kernel and the C ABI of include/wurblpt_hip.h.
 *
 * Kernel shape (MI355X first):
 *  - one lane = one pixel, because the reference consumes ONE Prng per pixel serially over
 *    all of that pixel's samples (wurblpt.hpp:342-366): the only parallel axis is pixels.
 *  - a lane never idles between samples or path segments: the loop below is a per-lane state
 *    machine (NEW -> PATH ray -> optional NEE ray -> advance) with ONE traversal site, so
 *    the 64 lanes of a wave always meet again at the BVH traversal no matter how long each
 *    lane's path is.  Path rays and next-event rays share that traversal code.
 *  - BVH nodes are fetched as two dwordx4 per lane, triangles as three; the traversal stack
 *    is an LDS column per lane (conflict-free: one dword per lane per level), spilling to
 *    scratch only below level 32.
 *  - shading data (96 B per triangle) is read once per ray, after traversal.
 *  - no MFMA anywhere: there is no dense contraction on this path.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/wurblpt_hip.h"
#include "wpt_device.h"

using namespace wptd;

namespace {

constexpr int WG = 256;            /* threads per workgroup: 4 waves, one per SIMD */
constexpr int LDS_STACK_DEPTH = 32; /* levels kept in LDS; deeper levels spill to scratch */
constexpr int SPILL_DEPTH = 96;     /* 32 + 96 = the reference's 128-entry stack (bvh.hpp:230) */
constexpr uint32_t NO_HIT = 0xffffffffu;

struct KernelArgs {
    SceneView sv;
    wpt_camera cam;
    wpt_params par;
    uint32_t width, height, samplesSqrt;
    uint32_t blockStart, blockSize;
    float* frame;
    wpt_counters* counters;
};

struct LaneCounters {
    uint32_t rays, nodes, leaves, pdfs, scatters;
};

/* BVH::hit (bvh.hpp:277-311): unordered depth-first walk, left child first, closest hit wins,
 * a later candidate at equal distance replaces an earlier one. */
template<bool COUNT>
__device__ __forceinline__ Candidate traverse(const SceneView& sv, f3 org, f3 dir, float amin, float amax,
        uint32_t (*stack)[WG], LaneCounters& lc)
{
    const RayAux h = rayAux(dir);
    Candidate best;
    best.prim = NO_HIT;
    best.a = 0.0f;
    best.invDet = 0.0f;
    best.U = best.V = best.W = best.det = 0.0f;
    uint32_t spill[SPILL_DEPTH];
    uint32_t node = 0;
    int sp = 0;
    const int tid = threadIdx.x;
    if (COUNT)
        lc.rays++;
    for (;;) {
        const float4 n0 = sv.nodes[2 * (size_t)node];
        const float4 n1 = sv.nodes[2 * (size_t)node + 1];
        if (COUNT)
            lc.nodes++;
        bool descend = false;
        if (boxTest(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), org, h.inv, amin, amax)) {
            const uint32_t link = __float_as_uint(n1.z);
            const uint32_t kind = __float_as_uint(n1.w);
            if (kind == WPT_NODE_INNER) {
                if (sp < LDS_STACK_DEPTH)
                    stack[sp][tid] = link;
                else
                    spill[sp - LDS_STACK_DEPTH] = link;
                sp++;
                node++;
                descend = true;
            } else if (kind == WPT_NODE_TRIANGLE) {
                if (COUNT)
                    lc.leaves++;
                const float4 g0 = sv.triGeom[3 * (size_t)link + 0];
                const float4 g1 = sv.triGeom[3 * (size_t)link + 1];
                const float4 g2 = sv.triGeom[3 * (size_t)link + 2];
                Candidate c;
                if (triangleTest(mk3(g0.x, g0.y, g0.z), mk3(g1.x, g1.y, g1.z), mk3(g2.x, g2.y, g2.z), org, h, amin, amax, c)) {
                    c.prim = link;
                    best = c;
                    amax = c.a;
                }
            }
        }
        if (!descend) {
            if (sp == 0)
                break;
            sp--;
            node = sp < LDS_STACK_DEPTH ? stack[sp][tid] : spill[sp - LDS_STACK_DEPTH];
        }
    }
    return best;
}

/* HitableTriangle::pdfValue (hitable_triangle.hpp:405-423) for one hot spot */
__device__ __forceinline__ float hotSpotPdf(const SceneView& sv, uint32_t prim, f3 org, f3 dir, const RayAux& h)
{
    const float4 g0 = sv.triGeom[3 * (size_t)prim + 0];
    const float4 g1 = sv.triGeom[3 * (size_t)prim + 1];
    const float4 g2 = sv.triGeom[3 * (size_t)prim + 2];
    const f3 v0 = mk3(g0.x, g0.y, g0.z), v1 = mk3(g1.x, g1.y, g1.z), v2 = mk3(g2.x, g2.y, g2.z);
    Candidate c;
    float value = 0.0f;
    if (triangleTest(v0, v1, v2, org, h, 0.0f, k_maxval, c)) {
        f3 edgeCross = cross(sub(v1, v0), sub(v2, v0));
        float edgeCrossLength = __builtin_sqrtf(dot(edgeCross, edgeCross));
        f3 faceNormal = divs(edgeCross, edgeCrossLength);
        float faceArea = 0.5f * edgeCrossLength;
        float cosine = __builtin_fabsf(dot(faceNormal, neg(dir)));
        float distance_squared = c.a * c.a;
        value = distance_squared / (cosine * faceArea);
    }
    return value;
}

__device__ __forceinline__ float hotSpotsMeanPdf(const SceneView& sv, f3 org, f3 dir, float invCount, LaneCounters& lc, bool count)
{
    const RayAux h = rayAux(dir);
    float sum = 0.0f;
    for (uint32_t i = 0; i < sv.hotspotCount; i++) {
        sum += hotSpotPdf(sv, sv.hotspots[i].prim, org, dir, h);
        if (count)
            lc.pdfs++;
    }
    sum *= invCount;
    return sum;
}

/* SensorRGB::accumulateRadiance (sensor_rgb.hpp:63-80) */
__device__ __forceinline__ void accumulate(const wpt_params& par, f4 opl, float distanceToLight, f4 radiance, float& a0, float& a1, float& a2)
{
    const bool dOk = distanceToLight >= par.min_dist_to_light && distanceToLight <= par.max_dist_to_light;
    if (dOk && opl.x >= par.min_path_len && opl.x <= par.max_path_len)
        a0 += radiance.x;
    if (dOk && opl.y >= par.min_path_len && opl.y <= par.max_path_len)
        a1 += radiance.y;
    if (dOk && opl.z >= par.min_path_len && opl.z <= par.max_path_len)
        a2 += radiance.z;
}

enum { ST_NEW = 0, ST_PATH = 1, ST_NEE_LIGHT = 2, ST_NEE_ENV = 3 };

template<uint32_t F, bool COUNT>
__global__ __launch_bounds__(WG) void wpt_pathtrace(const KernelArgs args)
{
    __shared__ uint32_t stack[LDS_STACK_DEPTH][WG];

    const SceneView& sv = args.sv;
    const wpt_params& par = args.par;
    const uint32_t gid = blockIdx.x * WG + threadIdx.x;
    /* lanes beyond the block still run the loop zero times; no early return before LDS use */
    const bool inBlock = gid < args.blockSize;
    const uint32_t pixel = args.blockStart + (inBlock ? gid : 0);
    const uint32_t px = pixel % args.width;
    const uint32_t py = pixel / args.width;
    const uint32_t samples = inBlock ? args.samplesSqrt * args.samplesSqrt : 0;
    const float invSamplesSqrt = 1.0f / (float)args.samplesSqrt;
    const float invW = 1.0f / (float)args.width;
    const float invH = 1.0f / (float)args.height;
    const float invHotSpots = 1.0f / (float)sv.hotspotCount;
    const bool haveEnv = (F & FEAT_ENVMAP) && sv.envType != WPT_ENV_NONE;

    Prng prng;
    prngSeed(prng, pixel);
    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
    LaneCounters lc = { 0, 0, 0, 0, 0 };

    /* per-lane path state */
    uint32_t sampleIndex = 0;
    int state = ST_NEW;
    uint32_t pathComponent = 0;
    Ray ray;
    f4 att = mk4(1.0f, 1.0f, 1.0f, 1.0f);
    f4 opl = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    /* pending next-event state */
    f4 nextAtt = att, directAtt = att, srRi = att;
    f3 srDir = mk3(0.0f, 0.0f, 1.0f);
    float directPdf = 0.0f, neeWeight = 0.0f;
    uint32_t chosenPrim = NO_HIT;
    ray.o = mk3(0.0f, 0.0f, 0.0f);
    ray.d = mk3(0.0f, 0.0f, 1.0f);
    ray.ri = att;

    for (;;) {
        if (state == ST_NEW) {
            if (sampleIndex >= samples)
                break;
            /* wurblpt.hpp:349-360: stratified jitter, the vertical stratum is drawn first */
            float u = (float)px, v = (float)py;
            if (par.randomize_ray_over_pixel) {
                const uint32_t j = sampleIndex / args.samplesSqrt;
                const uint32_t i = sampleIndex % args.samplesSqrt;
                const float fj = (float)j + in01(prng);
                const float fi = (float)i + in01(prng);
                u += fi * invSamplesSqrt;
                v += fj * invSamplesSqrt;
            } else {
                u += 0.5f;
                v += 0.5f;
            }
            u *= invW;
            v *= invH;
            /* Camera::getRay (camera.hpp:123-185), pinhole or thin lens */
            f3 P = mk3(mixr(args.cam.l, args.cam.r, u), mixr(args.cam.b, args.cam.t, v), -1.0f);
            f3 O = mk3(0.0f, 0.0f, 0.0f);
            if ((F & FEAT_LENS) && args.cam.lens_radius > 0.0f) {
                P = sclr(P, args.cam.focus_dist);
                f2 d = inUnitDisk(in01x2(prng));
                O = mk3(args.cam.lens_radius * d.x, args.cam.lens_radius * d.y, 0.0f);
            }
            f3 D = sub(P, O);
            O = add(O, mk3(0.0f, 0.0f, 0.0f));
            ray.o = add(ld3(args.cam.translation), quatRotate(args.cam.rotation, mul(O, ld3(args.cam.scaling))));
            ray.d = normalize(quatRotate(args.cam.rotation, D));
            ray.ri = mk4(1.0f, 1.0f, 1.0f, 1.0f);
            att = mk4(1.0f, 1.0f, 1.0f, 1.0f);
            opl = mk4(0.0f, 0.0f, 0.0f, 0.0f);
            pathComponent = 0;
            sampleIndex++;
            state = ST_PATH;
        }

        /* the one traversal site: path rays and next-event rays */
        const Candidate cand = traverse<COUNT>(sv, ray.o, ray.d, par.min_hit_distance, k_maxval, stack, lc);

        bool advance = false;
        if (state == ST_PATH) {
            if (cand.prim == NO_HIT) {
                if (haveEnv) {
                    f4 rad = mul(att, envL(sv, ray.d));
                    accumulate(par, mk4(k_maxval, k_maxval, k_maxval, k_maxval), k_maxval, rad, acc0, acc1, acc2);
                }
                state = ST_NEW;
                continue;
            }
            opl = add(opl, scl(cand.a, ray.ri));
            if (!(pathComponent + 1 < par.max_path_components)) {
                state = ST_NEW;
                continue;
            }
            Hit h = finishHit(sv, cand, ray.o, ray.d);
            const wpt_material& m = resolveMaterial<F>(sv, h.material, h);
            if (COUNT)
                lc.scatters++;
            const Scatter sr = materialScatter<F>(sv, m, ray, h, prng);
            {
                f4 rad = mul(att, materialEmitted<F>(sv, m, h));
                accumulate(par, opl, (pathComponent == 0 ? 0.0f : h.a), rad, acc0, acc1, acc2);
            }
            if (sr.type == SCATTER_NONE) {
                state = ST_NEW;
                continue;
            }
            nextAtt = mul(att, sr.att);
            if (sr.type == SCATTER_RANDOM) {
                if (sr.pdf > 0.0f)
                    nextAtt = divs(nextAtt, sr.pdf);
                else
                    nextAtt = mk4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            srDir = sr.dir;
            srRi = sr.ri;
            advance = true;
            if (sr.type == SCATTER_RANDOM && sv.hotspotCount > 0) {
                /* light sampling with MIS (wurblpt.hpp:179-220) */
                const float hotSpotsPdf = hotSpotsMeanPdf(sv, h.p, sr.dir, invHotSpots, lc, COUNT);
                nextAtt = sclr(nextAtt, powerHeuristicWeight(sr.pdf, hotSpotsPdf));
                uint32_t idx = (uint32_t)(in01(prng) * (float)sv.hotspotCount);
                idx = idx < sv.hotspotCount - 1 ? idx : sv.hotspotCount - 1;
                const wpt_hotspot& hs = sv.hotspots[idx];
                /* HitableTriangle::direction (hitable_triangle.hpp:425-443) */
                const f3 bary = inTriangle(in01x2(prng));
                f3 p = add(add(scl(bary.x, ld3(hs.p0)), scl(bary.y, ld3(hs.p1))), scl(bary.z, ld3(hs.p2)));
                if (hs.transform)
                    p = mat4mulPoint(hs.M, p);
                const f3 directDir = normalize(sub(p, h.p));
                directPdf = hotSpotsMeanPdf(sv, h.p, directDir, invHotSpots, lc, COUNT);
                if (directPdf > 0.0f) {
                    float dpdf;
                    materialEval<F>(sv, m, ray, h, directDir, directAtt, dpdf);
                    if (dpdf > 0.0f) {
                        neeWeight = powerHeuristicWeight(directPdf, dpdf);
                        chosenPrim = hs.prim;
                        ray.o = h.p;
                        ray.d = directDir;
                        state = ST_NEE_LIGHT;
                        advance = false;
                    }
                }
            } else if ((F & FEAT_ENVMAP) && sr.type == SCATTER_RANDOM && haveEnv && sv.envN > 0) {
                /* environment sampling with MIS (wurblpt.hpp:221-252) */
                const float lightsP = envP(sv, sr.dir);
                nextAtt = sclr(nextAtt, powerHeuristicWeight(sr.pdf, lightsP));
                const f3 lightDir = envD(sv, prng);
                directPdf = envP(sv, lightDir);
                float dpdf;
                materialEval<F>(sv, m, ray, h, lightDir, directAtt, dpdf);
                if (dpdf > 0.0f) {
                    neeWeight = powerHeuristicWeight(directPdf, dpdf);
                    ray.o = h.p;
                    ray.d = lightDir;
                    state = ST_NEE_ENV;
                    advance = false;
                }
            }
            if (advance)
                ray.o = h.p;
        } else if (state == ST_NEE_LIGHT) {
            /* wurblpt.hpp:208-218: only the CHOSEN hot spot as nearest hit counts */
            if (cand.prim == chosenPrim) {
                Hit lh = finishHit(sv, cand, ray.o, ray.d);
                const wpt_material& lm = resolveMaterial<F>(sv, lh.material, lh);
                f4 rad = mul(sclr(divs(mul(att, directAtt), directPdf), neeWeight), materialEmitted<F>(sv, lm, lh));
                f4 oplLight = add(opl, scl(lh.a, ray.ri));
                accumulate(par, oplLight, lh.a, rad, acc0, acc1, acc2);
            }
            state = ST_PATH;
            advance = true;
        } else { /* ST_NEE_ENV */
            if (cand.prim == NO_HIT) {
                f4 rad = mul(sclr(divs(mul(att, directAtt), directPdf), neeWeight), envL(sv, ray.d));
                accumulate(par, mk4(k_maxval, k_maxval, k_maxval, k_maxval), k_maxval, rad, acc0, acc1, acc2);
            }
            state = ST_PATH;
            advance = true;
        }

        if (advance) {
            /* wurblpt.hpp:254-273 (ray.o already is the hit position) */
            att = nextAtt;
            ray.d = srDir;
            ray.ri = srRi;
            const float mx = max4(att);
            if (mx < par.rr_threshold && pathComponent >= 5) {
                const float q = clampr(1.0f - mx, 0.0f, 0.95f);
                if (in01(prng) < q) {
                    state = ST_NEW;
                    continue;
                }
                const float rrWeight = 1.0f / (1.0f - q);
                att = sclr(att, rrWeight);
            }
            pathComponent++;
        }
    }

    if (inBlock) {
        /* SensorRGB::finishPixel (sensor_rgb.hpp:82-87) */
        const float invSamples = 1.0f / (float)(args.samplesSqrt * args.samplesSqrt);
        float* out = args.frame + 3 * (size_t)pixel;
        out[0] = invSamples * acc0;
        out[1] = invSamples * acc1;
        out[2] = invSamples * acc2;
    }
    if (COUNT && args.counters) {
        atomicAdd((unsigned long long*)&args.counters->samples, (unsigned long long)samples);
        atomicAdd((unsigned long long*)&args.counters->rays, (unsigned long long)lc.rays);
        atomicAdd((unsigned long long*)&args.counters->node_visits, (unsigned long long)lc.nodes);
        atomicAdd((unsigned long long*)&args.counters->leaf_tests, (unsigned long long)lc.leaves);
        atomicAdd((unsigned long long*)&args.counters->pdf_tests, (unsigned long long)lc.pdfs);
        atomicAdd((unsigned long long*)&args.counters->scatters, (unsigned long long)lc.scatters);
    }
}

/* Bit-parity self test of the arithmetic the kernel relies on: ops 0..5 are the
 * transcendentals of wpt_math.h, 6 = IEEE division, 7 = IEEE square root. */
__global__ void wpt_selftest_kernel(int op, int n, const float* a, const float* b, float* out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    float x = a[i], y = b[i], r;
    switch (op) {
    case 0: r = wptm::sinf_(x); break;
    case 1: r = wptm::cosf_(x); break;
    case 2: r = wptm::expf_(x); break;
    case 3: r = wptm::powf_(x, y); break;
    case 4: r = wptm::asinf_(x); break;
    case 5: r = wptm::atan2f_(x, y); break;
    case 6: r = x / y; break;
    case 7: r = __builtin_sqrtf(x); break;
    case 8: r = x * y + x; break; /* must stay unfused */
    default: r = 1.0f / x; break;
    }
    out[i] = r;
}

/* per-bin importance of the environment map (envmap.hpp:128-140) */
__global__ void wpt_env_importance_kernel(SceneView sv, int N, float* importance)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * N)
        return;
    int x = i % N, y = i / N;
    f2 uv;
    uv.y = ((float)y + 0.5f) / (float)N;
    uv.x = ((float)x + 0.5f) / (float)N;
    f4 L = envL(sv, envInvM(uv));
    importance[i] = L.x + L.y + L.z + L.w;
}

/* ---- host side of the C ABI ---- */

thread_local std::string g_error;

wpt_status fail(wpt_status s, const std::string& msg)
{
    g_error = msg;
    return s;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(WPT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

} /* namespace */

struct wpt_scene {
    int device;
    SceneView view;
    uint32_t features;
    uint32_t nodeCount, triCount;
    std::vector<void*> allocations;
    std::vector<float> envM, envMcs;
    std::vector<int32_t> envMs;
};

namespace {

uint32_t g_threadsPerGroup = WG;
uint32_t g_variant = 0;

template<typename T> wpt_status uploadArray(wpt_scene* s, const T* src, size_t count, const T** dst)
{
    *dst = nullptr;
    size_t bytes = count * sizeof(T);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes > 0 ? bytes : 16);
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? WPT_ERR_OUT_OF_MEMORY : WPT_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    s->allocations.push_back(p);
    if (bytes > 0)
        HIP_TRY(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    *dst = static_cast<const T*>(p);
    return WPT_OK;
}

uint32_t sceneFeatures(const wpt_scene_desc* d)
{
    uint32_t f = 0;
    for (uint32_t i = 0; i < d->material_count; i++) {
        const wpt_material& m = d->materials[i];
        if (m.type == WPT_MAT_MODPHONG)
            f |= FEAT_MODPHONG;
        if (m.type == WPT_MAT_TWOSIDED)
            f |= FEAT_TWOSIDED;
        if (m.type == WPT_MAT_GGX)
            f |= FEAT_GGX;
        if (m.type == WPT_MAT_GLASS || m.type == WPT_MAT_MIRROR)
            f |= FEAT_GLASS;
        bool tex = m.normal_tex >= 0;
        if (m.type != WPT_MAT_TWOSIDED)
            for (int k = 0; k < 5; k++)
                tex = tex || m.tex[k] >= 0;
        if (tex)
            f |= FEAT_TEXTURES;
    }
    if (d->envmap.type != WPT_ENV_NONE)
        f |= FEAT_ENVMAP | FEAT_TEXTURES;
    return f;
}

wpt_status validate(const wpt_scene_desc* d)
{
    if (!d)
        return fail(WPT_ERR_INVALID_ARGUMENT, "scene description is NULL");
    if (d->abi_version != WPT_ABI_VERSION)
        return fail(WPT_ERR_INVALID_ARGUMENT, "scene description has a different ABI version");
    if (d->node_count == 0 || !d->nodes)
        return fail(WPT_ERR_INVALID_ARGUMENT, "scene has no BVH nodes (run Scene::updateBVH)");
    /* every index the kernel will follow must stay inside its array: a bad index would be an
     * out-of-bounds access on the GPU */
    for (uint32_t i = 0; i < d->node_count; i++) {
        const wpt_bvh_node& n = d->nodes[i];
        if (n.kind == WPT_NODE_INNER) {
            if (n.link >= d->node_count || n.link <= i || i + 1 >= d->node_count)
                return fail(WPT_ERR_INVALID_ARGUMENT, "BVH inner node links outside the node array");
        } else if (n.kind == WPT_NODE_TRIANGLE) {
            if (n.link >= d->tri_count)
                return fail(WPT_ERR_INVALID_ARGUMENT, "BVH leaf references a triangle outside the array");
        } else if (n.kind != WPT_NODE_EMPTY) {
            return fail(WPT_ERR_UNSUPPORTED, "BVH node kind is not known to the kernel");
        }
    }
    for (uint32_t i = 0; i < d->tri_count; i++) {
        if (d->tri_geom[i].instance >= d->instance_count || d->tri_geom[i].material >= d->material_count)
            return fail(WPT_ERR_INVALID_ARGUMENT, "triangle references an instance or material outside the arrays");
    }
    for (uint32_t i = 0; i < d->material_count; i++) {
        const wpt_material& m = d->materials[i];
        if (m.type > WPT_MAT_TWOSIDED)
            return fail(WPT_ERR_UNSUPPORTED, "material type is not known to the kernel");
        if (m.type == WPT_MAT_TWOSIDED) {
            if (m.tex[0] < 0 || m.tex[1] < 0 || uint32_t(m.tex[0]) >= d->material_count || uint32_t(m.tex[1]) >= d->material_count)
                return fail(WPT_ERR_INVALID_ARGUMENT, "two-sided material references a material outside the array");
        } else {
            for (int k = 0; k < 5; k++)
                if (m.tex[k] >= int32_t(d->texture_count))
                    return fail(WPT_ERR_INVALID_ARGUMENT, "material references a texture outside the array");
        }
        if (m.normal_tex >= int32_t(d->texture_count))
            return fail(WPT_ERR_INVALID_ARGUMENT, "material references a normal map outside the array");
    }
    for (uint32_t i = 0; i < d->texture_count; i++) {
        const wpt_texture& t = d->textures[i];
        if (t.type > WPT_TEX_TRANSFORMER)
            return fail(WPT_ERR_UNSUPPORTED, "texture type is not known to the kernel");
        if (t.type == WPT_TEX_TRANSFORMER && (t.child < 0 || uint32_t(t.child) >= d->texture_count || uint32_t(t.child) >= i))
            return fail(WPT_ERR_INVALID_ARGUMENT, "texture transformer references a texture outside the array");
        if (t.type == WPT_TEX_IMAGE) {
            size_t cs = t.texel_type == WPT_TEXEL_U8 ? 1 : t.texel_type == WPT_TEXEL_U16 ? 2 : 4;
            if (t.width == 0 || t.height == 0 || t.comps < 1 || t.comps > 4 || t.texel_type > WPT_TEXEL_F32
                    || t.texel_offset + size_t(t.width) * t.height * t.comps * cs > d->texel_bytes)
                return fail(WPT_ERR_INVALID_ARGUMENT, "image texture lies outside the texel pool");
        }
    }
    for (uint32_t i = 0; i < d->hotspot_count; i++)
        if (d->hotspots[i].prim >= d->tri_count)
            return fail(WPT_ERR_INVALID_ARGUMENT, "hot spot references a triangle outside the array");
    if (d->envmap.type > WPT_ENV_EQUIRECT)
        return fail(WPT_ERR_UNSUPPORTED, "environment map type is not known to the kernel");
    if (d->envmap.type != WPT_ENV_NONE && (d->envmap.tex < 0 || uint32_t(d->envmap.tex) >= d->texture_count))
        return fail(WPT_ERR_INVALID_ARGUMENT, "environment map references a texture outside the array");
    return WPT_OK;
}

template<uint32_t F> void launchVariant(const KernelArgs& args, bool count, dim3 grid, hipStream_t stream)
{
    if (count)
        hipLaunchKernelGGL((wpt_pathtrace<F, true>), grid, dim3(WG), 0, stream, args);
    else
        hipLaunchKernelGGL((wpt_pathtrace<F, false>), grid, dim3(WG), 0, stream, args);
}

constexpr uint32_t FEAT_BASIC = FEAT_GGX | FEAT_GLASS;
constexpr uint32_t FEAT_ALL = FEAT_TEXTURES | FEAT_MODPHONG | FEAT_ENVMAP | FEAT_LENS | FEAT_TWOSIDED | FEAT_GGX | FEAT_GLASS;

} /* namespace */

extern "C" {

int wpt_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        g_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
        return 0;
    }
    return n;
}

wpt_status wpt_select_device(int device)
{
    HIP_TRY(hipSetDevice(device));
    return WPT_OK;
}

wpt_status wpt_scene_upload(const wpt_scene_desc* desc, wpt_scene** out_scene)
{
    if (!out_scene)
        return fail(WPT_ERR_INVALID_ARGUMENT, "out_scene is NULL");
    *out_scene = nullptr;
    wpt_status st = validate(desc);
    if (st != WPT_OK)
        return st;
    if (wpt_device_count() <= 0)
        return fail(WPT_ERR_NO_DEVICE, "no HIP device is available; the path tracer has no CPU fallback");
    wpt_scene* s = new wpt_scene;
    HIP_TRY(hipGetDevice(&s->device));
    s->features = sceneFeatures(desc);
    s->nodeCount = desc->node_count;
    s->triCount = desc->tri_count;
    memset(&s->view, 0, sizeof(s->view));
    const float4* nodes = nullptr;
    const float4* geom = nullptr;
    const float4* attr = nullptr;
#define UP(call)                  \
    do {                          \
        st = (call);              \
        if (st != WPT_OK) {       \
            wpt_scene_free(s);    \
            return st;            \
        }                         \
    } while (0)
    UP(uploadArray(s, reinterpret_cast<const float4*>(desc->nodes), size_t(desc->node_count) * 2, &nodes));
    UP(uploadArray(s, reinterpret_cast<const float4*>(desc->tri_geom), size_t(desc->tri_count) * 3, &geom));
    UP(uploadArray(s, reinterpret_cast<const float4*>(desc->tri_attr), size_t(desc->tri_count) * 6, &attr));
    s->view.nodes = nodes;
    s->view.triGeom = geom;
    s->view.triAttr = attr;
    UP(uploadArray(s, desc->instances, desc->instance_count, &s->view.instances));
    UP(uploadArray(s, desc->materials, desc->material_count, &s->view.materials));
    UP(uploadArray(s, desc->textures, desc->texture_count, &s->view.textures));
    UP(uploadArray(s, desc->texels, desc->texel_bytes, &s->view.texels));
    UP(uploadArray(s, desc->hotspots, desc->hotspot_count, &s->view.hotspots));
    s->view.hotspotCount = desc->hotspot_count;
    s->view.envType = desc->envmap.type;
    s->view.envCompat = desc->envmap.compat;
    s->view.envTex = desc->envmap.tex;
    s->view.envN = 0;
    if (desc->envmap.type != WPT_ENV_NONE && desc->envmap.N > 0) {
        const int N = desc->envmap.N;
        const size_t bins = size_t(N) * N;
        if (desc->envmap.M && desc->envmap.Ms && desc->envmap.Mcs) {
            s->envM.assign(desc->envmap.M, desc->envmap.M + bins);
            s->envMs.assign(desc->envmap.Ms, desc->envmap.Ms + bins);
            s->envMcs.assign(desc->envmap.Mcs, desc->envmap.Mcs + bins);
        } else {
            /* EnvironmentMap::initializeImportanceSampling (envmap.hpp:121-158): the per-bin
             * importance comes from the device's own L(); sum, sort and prefix sum run on the
             * host in the reference's sequential order */
            float* dImp = nullptr;
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&dImp), bins * sizeof(float));
            if (e != hipSuccess) {
                wpt_scene_free(s);
                return fail(WPT_ERR_OUT_OF_MEMORY, "hipMalloc for the importance map failed");
            }
            hipLaunchKernelGGL(wpt_env_importance_kernel, dim3((bins + 255) / 256), dim3(256), 0, 0, s->view, N, dImp);
            s->envM.resize(bins);
            e = hipMemcpy(s->envM.data(), dImp, bins * sizeof(float), hipMemcpyDeviceToHost);
            hipFree(dImp);
            if (e != hipSuccess) {
                wpt_scene_free(s);
                return fail(WPT_ERR_HIP, std::string("importance map: ") + hipGetErrorString(e));
            }
            float total = 0.0f;
            for (size_t i = 0; i < bins; i++)
                total += s->envM[i];
            for (size_t i = 0; i < bins; i++)
                s->envM[i] /= total;
            s->envMs.resize(bins);
            for (size_t i = 0; i < bins; i++)
                s->envMs[i] = int32_t(i);
            const std::vector<float>& M = s->envM;
            std::sort(s->envMs.begin(), s->envMs.end(), [&M](unsigned int i, unsigned int j) { return M[i] > M[j]; });
            s->envMcs.resize(bins);
            float sum = 0.0f;
            for (size_t i = 0; i < bins; i++) {
                sum += M[s->envMs[i]];
                s->envMcs[i] = sum;
            }
        }
        UP(uploadArray(s, s->envM.data(), bins, &s->view.envM));
        UP(uploadArray(s, s->envMs.data(), bins, &s->view.envMs));
        UP(uploadArray(s, s->envMcs.data(), bins, &s->view.envMcs));
        s->view.envN = N;
    }
#undef UP
    *out_scene = s;
    return WPT_OK;
}

void wpt_scene_free(wpt_scene* scene)
{
    if (!scene)
        return;
    for (void* p : scene->allocations)
        (void)hipFree(p);
    delete scene;
}

/* copies the importance tables of an uploaded scene back (tests compare them with the oracle's) */
wpt_status wpt_scene_get_envmap_tables(const wpt_scene* scene, float* M, int32_t* Ms, float* Mcs)
{
    if (!scene || scene->view.envN <= 0)
        return fail(WPT_ERR_INVALID_ARGUMENT, "scene has no importance tables");
    size_t bins = scene->envM.size();
    memcpy(M, scene->envM.data(), bins * sizeof(float));
    memcpy(Ms, scene->envMs.data(), bins * sizeof(int32_t));
    memcpy(Mcs, scene->envMcs.data(), bins * sizeof(float));
    return WPT_OK;
}

wpt_status wpt_render_block_device(wpt_scene* scene, const wpt_camera* camera, const wpt_params* params,
        uint32_t width, uint32_t height, uint32_t samples_sqrt, uint32_t block_start, uint32_t block_size,
        float* frame_device, wpt_counters* counters_device, void* hip_stream)
{
    if (!scene || !camera || !params || !frame_device)
        return fail(WPT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (width == 0 || height == 0 || samples_sqrt == 0 || samples_sqrt > 65535)
        return fail(WPT_ERR_INVALID_ARGUMENT, "width, height and samples_sqrt must be positive");
    if (uint64_t(width) * height > 0xffffffffull || uint64_t(block_start) + block_size > uint64_t(width) * height)
        return fail(WPT_ERR_INVALID_ARGUMENT, "pixel block lies outside the frame");
    if (block_size == 0)
        return WPT_OK;
    KernelArgs args;
    args.sv = scene->view;
    args.cam = *camera;
    args.par = *params;
    args.width = width;
    args.height = height;
    args.samplesSqrt = samples_sqrt;
    args.blockStart = block_start;
    args.blockSize = block_size;
    args.frame = frame_device;
    args.counters = counters_device;
    uint32_t need = scene->features | (camera->lens_radius > 0.0f ? FEAT_LENS : 0u);
    dim3 grid((block_size + WG - 1) / WG);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const bool count = counters_device != nullptr;
    if ((need & ~FEAT_BASIC) == 0 && g_variant != 2)
        launchVariant<FEAT_BASIC>(args, count, grid, stream);
    else
        launchVariant<FEAT_ALL>(args, count, grid, stream);
    HIP_TRY(hipGetLastError());
    return WPT_OK;
}

wpt_status wpt_render_block(wpt_scene* scene, const wpt_camera* camera, const wpt_params* params, uint32_t width,
        uint32_t height, uint32_t samples_sqrt, uint32_t block_start, uint32_t block_size, float* block_rgb)
{
    if (!block_rgb)
        return fail(WPT_ERR_INVALID_ARGUMENT, "block_rgb is NULL");
    if (block_size == 0)
        return WPT_OK;
    /* a frame-sized address space would waste memory for small blocks: allocate the block only
     * and bias the frame pointer so that pixel `block_start` lands at offset 0 */
    float* dBlock = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dBlock), size_t(block_size) * 3 * sizeof(float)));
    float* biased = dBlock - size_t(block_start) * 3;
    wpt_status st = wpt_render_block_device(scene, camera, params, width, height, samples_sqrt, block_start, block_size,
            biased, nullptr, nullptr);
    if (st == WPT_OK) {
        hipError_t e = hipMemcpy(block_rgb, dBlock, size_t(block_size) * 3 * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            st = fail(WPT_ERR_HIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
    }
    (void)hipFree(dBlock);
    return st;
}

wpt_status wpt_set_launch_config(uint32_t threads_per_group, uint32_t variant)
{
    if (threads_per_group != 0 && threads_per_group != WG)
        return fail(WPT_ERR_UNSUPPORTED, "this build uses 256 threads per workgroup");
    g_threadsPerGroup = WG;
    g_variant = variant;
    return WPT_OK;
}

const char* wpt_kernel_name(void)
{
    return "wpt_pathtrace";
}

const char* wpt_last_error(void)
{
    return g_error.c_str();
}

/* test hook: evaluates one arithmetic primitive on the device for n inputs (device pointers) */
wpt_status wpt_selftest_math(int op, int n, const float* a_device, const float* b_device, float* out_device)
{
    if (n <= 0)
        return WPT_OK;
    hipLaunchKernelGGL(wpt_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, op, n, a_device, b_device, out_device);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return WPT_OK;
}

} /* extern "C" */
