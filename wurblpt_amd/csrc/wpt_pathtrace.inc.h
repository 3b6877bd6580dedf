/*
 * wpt_pathtrace.inc.h -- the path-tracing kernel template (see wpt_capi.hip for the design
 * notes).  Included by one translation unit per instantiation (wpt_k_*.hip) so that the
 * variants compile in parallel.
 */
#ifndef WPT_PATHTRACE_INC_H
#define WPT_PATHTRACE_INC_H

#include <hip/hip_runtime.h>

#include "../../include/wurblpt_hip.h"
#include "wpt_device.h"

namespace wptk {

using namespace wptd;

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


constexpr uint32_t FEAT_BASIC = FEAT_GGX | FEAT_GLASS;
constexpr uint32_t FEAT_ALL = FEAT_TEXTURES | FEAT_MODPHONG | FEAT_ENVMAP | FEAT_LENS | FEAT_TWOSIDED | FEAT_GGX | FEAT_GLASS;

/* one launcher per instantiation, each defined in its own translation unit */
void launchBasic(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchBasicCount(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFull(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFullCount(const KernelArgs& args, dim3 grid, hipStream_t stream);

} /* namespace wptk */

#endif
