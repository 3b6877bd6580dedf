/*
 * wpt_wavefront.inc.h -- the wavefront form of the path tracer, for scenes that are fetched from HBM / L2.
 *
 * The single kernel of wpt_pathtrace.inc.h keeps a pixel in one lane through every stage, so its waves run node steps,
 * leaf tests and the long shading round with a fraction of their lanes each, at the register budget of the largest
 * stage (128 registers, four waves per SIMD).  Here the same path logic (wpt_blocks.h, unchanged) and the same walk
 * (same visiting order as bvh.hpp:277-311) are two kernels that hand a pixel's ray back and forth through HBM:
 *
 *   wf_trace   persistent waves (76 registers, built for WF_TRACE_WAVES waves per SIMD) that hold nothing but rays: origin, reciprocals,
 *              shear, node, bound, candidate.  A wave takes rays from the iteration's queue in chunks (one atomic per
 *              chunk) and deals them to its lanes as they finish, so its lanes are always walking; a finished lane
 *              writes the candidate to the pixel's record and files the pixel under the kind of shading it needs next
 *              (per-wave staging in LDS, one atomic per flush of up to 64 pixels).
 *   wf_shade   one lane per filed pixel, a wave per kind: the pixel's cold words are copied from its record into the LDS
 *              slots the blocks of wpt_blocks.h work on (the layout of the single kernel's long round), one path
 *              component is evaluated (blockNeeResult for the light ray that travelled beside the path ray, blockShade, then
 *              blockNew where the path ended), record and next rays are written back and the pixel is queued for the next trace
 *              (one atomic per workgroup).
 *
 * A pixel draws from its one generator in the reference's order (prng.hpp:79-101, wurblpt.hpp:342-366) and its accumulator receives the
 * reference's additions in the reference's order: a frame is the same bit for bit as the single kernel's and the oracle's.  An iteration
 * is trace + shade; the number of iterations is the largest number of path rays any pixel of the launch traces.  The launch's lanes may be
 * cut into groups that iterate on streams of their own (wpt_set_wavefront; one group by default).
 *
 * Per lane of the launch: one record of 16 quadwords (256 B) in HBM --
 *   0..7   the cold words of wpt_blocks.h (generator, attenuations, accumulator, ...), one 128-byte line
 *   8      the path ray's origin, time              9    its direction, the pixel's state word (WF_W_*)
 *   10     its candidate: primitive, distance, 1 / det, U          11   V, W | the light ray's candidate: V, W
 *   12     the light ray's candidate: primitive, distance, 1 / det, U
 *   13     the light ray's origin | a suspended walk's next node     14   its direction | a suspended walk's bound
 *   15     free
 *
 * Two rays per pixel and iteration (round 4).  A hit's light ray and the path's continuation are traced in the SAME iteration:
 * what follows a light ray's end in the reference -- Russian roulette, continuation -- does not depend on its answer
 * (wpt_blocks.h, blockShade<MERGED>), so the shade kernel sets both rays up, the trace kernel walks them one after the other
 * in one lane (light ray first), and the next shading adds the light ray's answer before it does anything else
 * (blockNeeResult): the accumulator sees the reference's additions in the reference's order.  Half the iterations, one
 * round trip of a pixel's record per path component instead of two.
 *
 * The walk of a light ray towards the environment ends at its first accepted hit, as in the single kernel (the answer it is
 * traced for -- anything in the way? -- is known there).
 *
 * A launch of the trace walks a ray for at most WfArgs::stepBudget node steps.  The walk has no stack: what it needs to
 * go on is the next node, the bound and the candidate, eight words.  A ray that is not through by then is written
 * back with them and queued for the next iteration's trace, which takes it up where it stopped -- same visiting order,
 * same result; its pixel simply is not shaded in between.  Without the budget every iteration lasts as long as its
 * longest ray (the trees here are 50 levels deep and ray lengths have a long tail): measured, the waves of a trace
 * launch were on the device for 53 % of its duration on average.
 */
#ifndef WPT_WAVEFRONT_INC_H
#define WPT_WAVEFRONT_INC_H

#include "wpt_pathtrace.inc.h"

namespace wptk {

constexpr uint32_t WF_SLOTS = 16; /* quadwords per lane record */
/* Waves per SIMD the trace is built for.  More waves do not walk faster: dependent fetches of random nodes reach their
 * highest rate at two to four waves per SIMD and fall off beyond (tools/micro/node_fetch.hip: 262 G fetches per second at
 * four, 137 G at eight for 17 MB of nodes) -- the lines a compute unit's lanes have in flight outgrow its L1. */
constexpr int WF_TRACE_WAVES = 4;
enum { WF_RAY_O = 8, WF_RAY_D = 9, WF_HIT0 = 10, WF_HIT1 = 11, WF_NEE_HIT = 12, WF_NEE_O = 13, WF_NEE_D = 14 };
/* the pixel's state word (record quadword 9, w) */
constexpr uint32_t WF_W_PATH = 1u;        /* a path ray is to be traced (or has been: its candidate is in the record) */
constexpr uint32_t WF_W_NEE_SHIFT = 1;    /* bits 1-2: the light ray beside it: 0 none, RAY_NEE_LIGHT, RAY_NEE_ENV */
constexpr uint32_t WF_W_NEE_MASK = 3u << WF_W_NEE_SHIFT;
constexpr uint32_t WF_W_FINISH = 8u;      /* the pixel's samples are through; its last light ray's answer is still to be added */
constexpr uint32_t WF_RESUME = 0x100u;    /* the record holds a suspended walk ... */
constexpr uint32_t WF_RESUME_PATH = 0x200u; /* ... of the path ray (the light ray is through), else of the light ray */

/* kinds a traced pixel is filed under for shading; a wave of wf_shade serves one kind */
enum {
    WF_B_MISS = 0,     /* path ray left the scene: environment radiance, next sample */
    WF_B_NEE = 1,      /* no path ray: the pixel's last light ray came back */
    WF_B_LIGHT = 2,    /* path ray on an emitter: emission, path ends */
    WF_B_LAMBERT = 3,
    WF_B_MODPHONG = 4,
    WF_B_GGX = 5,
    WF_B_EXPLICIT = 6, /* glass, mirror */
    WF_B_RGL = 7,
    WF_BUCKETS = 8
};

/* counters of one iteration; a ring of them, cleared ahead by the host */
struct WfIter {
    uint32_t rayCount;    /* pixels queued for this iteration's trace (appended by the shade before it) */
    uint32_t traceCursor; /* next queue entry to hand to a wave */
    uint32_t pad[2];
    uint32_t bucketCount[WF_BUCKETS]; /* traced pixels by kind (appended by the trace) */
};
constexpr uint32_t WF_RING = 256; /* ring entries, a power of two; the host clears one batch of iterations ahead */

struct WfArgs {
    KernelArgs k;       /* scene, camera, parameters, frame, lane -> pixel mapping of the launch */
    float4* state;      /* WF_SLOTS quadwords per lane of the launch */
    uint32_t* rayQueue[2];    /* lane indices queued for the trace of even / odd iterations (the group's own arrays) */
    uint32_t* bucketQueue;    /* WF_BUCKETS arrays of laneCount entries: traced lanes by kind */
    WfIter* ring;
    uint32_t laneFirst, laneCount; /* this group's lanes of the launch */
    uint32_t iteration;
    uint32_t buckets;    /* 1: the trace files pixels by kind and the shade takes them from there; 0: the shade walks the ray queue */
    uint32_t chunk;      /* queue entries a wave of the trace takes per atomic */
    uint32_t refillIdle; /* the trace deals new rays once this many lanes of a wave have finished */
    uint32_t leafBias;
    uint32_t stepBudget; /* node steps a ray may take per launch of the trace (0: no limit) */
    uint32_t topNodes;   /* the first topNodes nodes of the array (the top of a large tree, stored level by level) are walked from LDS */
    uint32_t kindMask;   /* shade: the kinds this launch serves (measurements launch one per kind; normally all) */
};

WPT_D uint32_t laneId() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
WPT_D uint32_t rankIn(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

/* ---- trace ---- */
enum { T_NODE = 0, T_LEAF = 1, T_DONE = 2, T_IDLE = 3, T_SUSPEND = 4 };

/* kind of shading a traced ray needs (TwoSided resolved as resolveMaterial does, material.hpp:273-320).  Only the
 * grouping of lanes into waves depends on it, never a value. */
template<bool SPHERES> WPT_D uint32_t shadeKind(const SceneView& sv, uint32_t word, uint32_t prim, float invDet)
{
    if (!(word & WF_W_PATH))
        return WF_B_NEE; /* nothing to shade: a last light ray's answer, then the pixel is written */
    if (prim == NO_HIT)
        return WF_B_MISS;
    uint32_t mat;
    const bool backside = invDet < 0.0f; /* spheres: the hit record decides; a wrong guess only files the pixel elsewhere */
    if (SPHERES && (prim & PRIM_SPHERE))
        mat = sv.spheres[prim & ~PRIM_SPHERE].material;
    else
        mat = __float_as_uint(reinterpret_cast<const float*>(sv.triGeom + 3 * (size_t)prim + 1)[3]);
    const wpt_material* m = sv.materials + mat;
    for (int guard = 0; guard < 4 && m->type == WPT_MAT_TWOSIDED; guard++)
        m = sv.materials + (backside ? m->tex[1] : m->tex[0]);
    switch (m->type) {
    case WPT_MAT_LIGHT_DIFFUSE: return WF_B_LIGHT;
    case WPT_MAT_LAMBERTIAN: return WF_B_LAMBERT;
    case WPT_MAT_MODPHONG: return WF_B_MODPHONG;
    case WPT_MAT_GGX: return WF_B_GGX;
    case WPT_MAT_RGL: return WF_B_RGL;
    default: return WF_B_EXPLICIT;
    }
}

template<bool SPHERES>
__global__ __launch_bounds__(WG, WF_TRACE_WAVES) void wf_trace(const WfArgs a)
{
    const SceneView& sv = a.k.sv;
    WfIter* const cur = a.ring + (a.iteration & (WF_RING - 1));
    WfIter* const nxt = a.ring + ((a.iteration + 1) & (WF_RING - 1));
    uint32_t* const queueNext = a.rayQueue[(a.iteration + 1) & 1u];
    const uint32_t count = cur->rayCount;
    const uint32_t nodeCount = sv.nodeCount;
    const float amin = a.k.par.min_hit_distance;
    /* The top of the tree in LDS: every ray starts there, and with thousands of lanes per compute unit in flight L1 does
     * not keep those lines, so the same few lines would be asked of the same few L2 channels by every compute unit. */
    extern __shared__ float4 ldsTop[];
    const uint32_t topNodes = count != 0 ? a.topNodes : 0u;
    for (uint32_t i = threadIdx.x; i < 2 * topNodes; i += WG)
        ldsTop[i] = sv.nodes[i];
    __syncthreads();
    /* fetch of the node a lane tests next, behind the step that found it */
    auto fetchNode = [&](uint32_t node, bool want, float4& pn0, float4& pn1) {
        if (!want)
            return;
        if (node < topNodes) {
            pn0 = ldsTop[2 * node];
            pn1 = ldsTop[2 * node + 1];
        } else {
            pn0 = sv.nodes[2 * (size_t)node];
            pn1 = sv.nodes[2 * (size_t)node + 1];
        }
    };
    const uint32_t* const queue = a.rayQueue[a.iteration & 1u];
    const uint32_t lane = laneId();
    /* filed pixels are staged per wave and kind in LDS (64 entries each) and flushed to the kind's queue with one atomic */
    __shared__ uint32_t stage[WG / 64][WF_BUCKETS][64];
    uint32_t (*const myStage)[64] = stage[threadIdx.x >> 6];
    unsigned long long fills = 0; /* wave-uniform: eight 8-bit fill counts */
    auto flush = [&](uint32_t kd, uint32_t n) {
        uint32_t base = 0;
        if (lane == 0)
            base = atomicAdd(&cur->bucketCount[kd], n);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (lane < n)
            a.bucketQueue[(size_t)kd * a.laneCount + base + lane] = myStage[kd][lane];
        __builtin_amdgcn_wave_barrier();
    };

    uint32_t chunkNext = 0, chunkEnd = 0; /* wave-uniform: this wave's entries of the queue */
    bool dry = false;
    int state = T_IDLE;
    f3 o = mk3(0.0f, 0.0f, 0.0f), d = o;
    RayAux aux;
    aux.inv = o;
    aux.k = 0;
    aux.Sx = aux.Sy = 0.0f;
    uint32_t node = 0, leafPrim = 0, gid = 0, stepsLeft = 0;
    bool shadowRay = false;
    float amax = k_maxval;
    Candidate best;
    best.prim = NO_HIT;
    best.a = best.invDet = best.U = best.V = best.W = 0.0f;
    float4 pn0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pn1 = pn0;
    bool pathPhase = true; /* the lane walks its pixel's path ray (else: the light ray beside it) */
    /* the lane starts the ray of its phase from the root */
    auto startRay = [&](const float4* rec, uint32_t word) {
        const float4 ro = rec[pathPhase ? WF_RAY_O : WF_NEE_O], rd = rec[pathPhase ? WF_RAY_D : WF_NEE_D];
        o = mk3(ro.x, ro.y, ro.z);
        d = mk3(rd.x, rd.y, rd.z);
        aux = rayAux(d);
        if (rayMayNan(o, aux.inv)) /* only such rays test their slab distances for NaN per box (wpt_device.h) */
            aux.k |= RAY_MAY_NAN;
        node = 0;
        amax = k_maxval;
        best.prim = NO_HIT;
        state = T_NODE;
        /* a light ray towards the environment: its walk ends at the first accepted hit (wpt_pathtrace.inc.h) */
        shadowRay = !pathPhase && a.k.shadowWalksEnd != 0 && ((word & WF_W_NEE_MASK) >> WF_W_NEE_SHIFT) == (uint32_t)RAY_NEE_ENV;
    };

    for (;;) {
        /* ---- deal rays to the lanes that have none ---- */
        const unsigned long long idle = __ballot(state == T_IDLE);
        if (idle != 0) {
            if (chunkNext == chunkEnd && !dry) {
                uint32_t first = 0;
                if (lane == 0)
                    first = atomicAdd(&cur->traceCursor, a.chunk);
                first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
                if (first >= count) {
                    dry = true; /* the cursor only grows: nothing more for this wave */
                } else {
                    chunkNext = first;
                    chunkEnd = first + a.chunk < count ? first + a.chunk : count;
                }
            }
            const uint32_t avail = chunkEnd - chunkNext, want = (uint32_t)__popcll(idle);
            const uint32_t take = avail < want ? avail : want;
            if (take != 0) {
                const uint32_t rank = rankIn(idle);
                bool dealt = false;
                if (state == T_IDLE && rank < take) {
                    dealt = true;
                    gid = queue[chunkNext + rank];
                    const float4* rec = a.state + (size_t)gid * WF_SLOTS;
                    const uint32_t word = __float_as_uint(rec[WF_RAY_D].w);
                    /* the light ray first (if there is one and it is not through yet), then the path ray */
                    pathPhase = (word & WF_RESUME) ? (word & WF_RESUME_PATH) != 0 : (word & WF_W_NEE_MASK) == 0;
                    startRay(rec, word);
                    if (word & WF_RESUME) { /* a suspended walk goes on */
                        const float4 h0 = rec[pathPhase ? WF_HIT0 : WF_NEE_HIT], h1 = rec[WF_HIT1];
                        best.prim = __float_as_uint(h0.x);
                        best.a = h0.y;
                        best.invDet = h0.z;
                        best.U = h0.w;
                        best.V = pathPhase ? h1.x : h1.z;
                        best.W = pathPhase ? h1.y : h1.w;
                        node = __float_as_uint(rec[WF_NEE_O].w);
                        amax = rec[WF_NEE_D].w;
                    }
                    stepsLeft = a.stepBudget ? a.stepBudget : 0xffffffffu;
                }
                fetchNode(node, dealt, pn0, pn1);
                chunkNext += take;
            }
        }
        if (__ballot(state != T_IDLE) == 0) {
            if (dry)
                break;
            continue; /* the chunk ran out: take the next one */
        }
        /* walk until refillIdle lanes have finished (while there is something to deal them), or to the end */
        const bool canRefill = chunkNext != chunkEnd || !dry;
        const bool nanPossible = sv.boxesMayBeNan != 0 || __ballot((aux.k & RAY_MAY_NAN) != 0 && state != T_IDLE) != 0; /* rays are dealt above only */
        int leaveBelow = canRefill ? 65 - (int)a.refillIdle : 1;
        leaveBelow = leaveBelow < 1 ? 1 : leaveBelow;
        for (;;) {
            const int nNode = __popcll(__ballot(state == T_NODE));
            const int nLeaf = __popcll(__ballot(state == T_LEAF));
            if (nNode + nLeaf < leaveBelow)
                break;
            if (nLeaf * (int)a.leafBias >= nNode * 8 && nLeaf > 0) {
                bool wantNode = false;
                if (state == T_LEAF) {
                    /* HitableTriangle::hit / HitableSphere::hit, candidate part; `node` already is the node to go on with */
                    Candidate c;
                    bool accepted;
                    if (SPHERES && (leafPrim & PRIM_SPHERE)) {
                        c.invDet = c.U = c.V = c.W = 0.0f;
                        accepted = sphereTest(sv.spheres[leafPrim & ~PRIM_SPHERE], o, d, amin, amax, c.a);
                    } else {
                        const float4* g = sv.triGeom + 3 * (size_t)leafPrim;
                        const float4 g0 = g[0], g1 = g[1], g2 = g[2];
                        accepted = triangleTest(mk3(g0.x, g0.y, g0.z), mk3(g1.x, g1.y, g1.z), mk3(g2.x, g2.y, g2.z), o, aux, amin, amax, c);
                    }
                    if (accepted) {
                        c.prim = leafPrim;
                        best = c;
                        amax = c.a;
                    }
                    state = (node >= nodeCount || (accepted && shadowRay)) ? (int)T_DONE : (int)T_NODE;
                    if (state == T_NODE && stepsLeft == 0)
                        state = T_SUSPEND; /* between two nodes: nothing pending but (node, bound, candidate) */
                    wantNode = state == T_NODE;
                }
                fetchNode(node, wantNode, pn0, pn1);
            } else {
                bool wantNode = false;
                if (state == T_NODE) {
                    /* AABB::mayHit + the stackless form of BVH::hit's walk (wpt_pathtrace.inc.h) */
                    const float4 n0 = pn0, n1 = pn1;
                    const uint32_t skip = __float_as_uint(n1.z);
                    const uint32_t word = __float_as_uint(n1.w);
                    bool hit = boxTest<false>(nodeLo(n0, n1), nodeHi(n0, n1), o, aux.inv, amin, amax);
                    if (nanPossible) /* wave-uniform, rarely true */
                        hit = boxTest<true>(nodeLo(n0, n1), nodeHi(n0, n1), o, aux.inv, amin, amax);
                    const bool inner = word >= NODE_CHILD;
                    const bool toLeaf = hit && !inner;
                    leafPrim = toLeaf ? word : leafPrim;
                    node = (hit && inner) ? (word & NODE_INDEX_MASK) : skip;
                    state = toLeaf ? (int)T_LEAF : (int)T_NODE;
                    if (!toLeaf && node >= nodeCount)
                        state = T_DONE;
                    stepsLeft = stepsLeft ? stepsLeft - 1 : 0;
                    if (state == T_NODE && stepsLeft == 0)
                        state = T_SUSPEND;
                    wantNode = state == T_NODE;
                }
                fetchNode(node, wantNode, pn0, pn1);
            }
        }
        /* ---- finished rays: the candidate goes to the pixel's record; a light ray's lane goes on with the pixel's path ray;
         * a pixel whose rays are through goes to the queue of its kind; rays out of steps are written back as they stand and
         * queued for the next trace ---- */
        if (__ballot(state == T_DONE || state == T_SUSPEND) != 0) {
            uint32_t kind = WF_BUCKETS;
            const bool suspend = state == T_SUSPEND;
            bool again = false; /* the lane has started its pixel's path ray */
            if (state == T_DONE || suspend) {
                float4* rec = a.state + (size_t)gid * WF_SLOTS;
                rec[pathPhase ? WF_HIT0 : WF_NEE_HIT] = make_float4(__uint_as_float(best.prim), best.a, best.invDet, best.U);
                float* vw = reinterpret_cast<float*>(rec + WF_HIT1) + (pathPhase ? 0 : 2);
                vw[0] = best.V;
                vw[1] = best.W;
                uint32_t* wordAt = reinterpret_cast<uint32_t*>(rec + WF_RAY_D) + 3;
                const uint32_t word = *wordAt; /* read again here: one register less while walking */
                const uint32_t clean = word & ~(WF_RESUME | WF_RESUME_PATH);
                if (suspend) {
                    reinterpret_cast<uint32_t*>(rec + WF_NEE_O)[3] = node;
                    reinterpret_cast<float*>(rec + WF_NEE_D)[3] = amax;
                    *wordAt = clean | WF_RESUME | (pathPhase ? WF_RESUME_PATH : 0u);
                    state = T_IDLE;
                } else if (!pathPhase && (word & WF_W_PATH)) {
                    if (word != clean)
                        *wordAt = clean;
                    pathPhase = true;
                    startRay(rec, clean);
                    again = true;
                } else {
                    if (word != clean)
                        *wordAt = clean; /* through: the record holds results only */
                    if (a.buckets)
                        kind = shadeKind<SPHERES>(sv, clean, best.prim, best.invDet);
                    state = T_IDLE;
                }
            }
            fetchNode(node, again, pn0, pn1);
            const unsigned long long suspended = __ballot(suspend);
            if (suspended != 0) {
                uint32_t base = 0;
                if (lane == 0)
                    base = atomicAdd(&nxt->rayCount, (uint32_t)__popcll(suspended));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (suspend)
                    queueNext[base + rankIn(suspended)] = gid;
            }
            unsigned long long left = __ballot(kind < WF_BUCKETS);
            while (left != 0) {
                const int leader = __ffsll((long long)left) - 1;
                const uint32_t kd = (uint32_t)__builtin_amdgcn_readlane((int)kind, leader);
                const unsigned long long mine = __ballot(kind == kd);
                left &= ~mine;
                const uint32_t n = (uint32_t)__popcll(mine);
                uint32_t f = (uint32_t)(fills >> (8 * kd)) & 0xffu;
                if (f + n > 64u) {
                    flush(kd, f);
                    f = 0;
                }
                if (kind == kd)
                    myStage[kd][f + rankIn(mine)] = gid;
                __builtin_amdgcn_wave_barrier();
                f += n;
                if (f == 64u) {
                    flush(kd, f);
                    f = 0;
                }
                fills = (fills & ~(0xffull << (8 * kd))) | ((unsigned long long)f << (8 * kd));
            }
        }
    }
    if (a.buckets) {
        for (uint32_t kd = 0; kd < WF_BUCKETS; kd++) {
            const uint32_t f = (uint32_t)(fills >> (8 * kd)) & 0xffu;
            if (f != 0)
                flush(kd, f);
        }
    }
}

/* ---- shade ---- */

/* Workgroup -> (kind, first entry) over the kinds' queues laid end to end, each kind rounded up to whole workgroups.
 * false: nothing for this workgroup. */
WPT_D bool shadeSlice(const WfIter* cur, uint32_t kindMask, uint32_t group, uint32_t& kind, uint32_t& first, uint32_t& count)
{
    uint32_t at = 0;
    for (uint32_t kd = 0; kd < WF_BUCKETS; kd++) {
        const uint32_t n = ((kindMask >> kd) & 1u) ? cur->bucketCount[kd] : 0u;
        const uint32_t groups = (n + WG - 1) / WG;
        if (group < at + groups) {
            kind = kd;
            first = (group - at) * WG;
            count = n;
            return true;
        }
        at += groups;
    }
    return false;
}

/* INIT: the first launch of a group -- every lane takes its pixel, seeds the generator and starts the first sample.
 * Otherwise one path component per queued pixel. */
#ifndef WF_SHADE_WAVES
#define WF_SHADE_WAVES 4 /* waves per SIMD the shade kernels are built for (their LDS allows four workgroups per compute unit) */
#endif
template<uint32_t F, bool INIT>
__global__ __launch_bounds__(WG, WF_SHADE_WAVES) void wf_shade(const WfArgs a)
{
    /* [ math tables ][ cold path words: SLOT_COUNT x WG float4 ][ 8 words: compaction ] */
    extern __shared__ float4 lds[];
    float4* const ldsCold = lds + TABLE_BYTES / 16;
    uint32_t* const ldsWords = reinterpret_cast<uint32_t*>(ldsCold + SLOT_COUNT * WG);
#if defined(WPT_MATH_TABLES_IN_LDS) && defined(__HIP_DEVICE_COMPILE__)
    wptm::tables_to_lds(threadIdx.x);
#endif
    const SceneView& sv = a.k.sv;
    const wpt_params& par = a.k.par;
    WfIter* const cur = a.ring + (a.iteration & (WF_RING - 1));
    WfIter* const nxt = a.ring + ((a.iteration + 1) & (WF_RING - 1));
    uint32_t* const queueOut = a.rayQueue[(a.iteration + 1) & 1u];

    /* which entry of which queue */
    uint32_t kind = WF_BUCKETS, first = blockIdx.x * WG, count = 0;
    const uint32_t* queueIn = nullptr;
    if (INIT) {
        count = a.laneCount;
    } else if (a.buckets) {
        if (shadeSlice(cur, a.kindMask, blockIdx.x, kind, first, count))
            queueIn = a.bucketQueue + (size_t)kind * a.laneCount;
        else
            return; /* uniform for the workgroup */
    } else {
        count = cur->rayCount;
        queueIn = a.rayQueue[a.iteration & 1u];
        if (first >= count)
            return;
    }
    __syncthreads(); /* the tables */

    FrameArgs fa;
    fa.cam = a.k.cam;
    fa.par = a.k.par;
    fa.width = a.k.width;
    fa.height = a.k.height;
    fa.samplesSqrt = a.k.samplesSqrt;
    fa.invWidth = a.k.invWidth;
    fa.invHeight = a.k.invHeight;
    fa.invSamplesSqrt = a.k.invSamplesSqrt;
    TriGeomFromScene tri4;
    tri4.triGeom = sv.triGeom;

    PathLds<WG> ps;
    ps.base = ldsCold + threadIdx.x;
    const uint32_t entry = first + threadIdx.x;
    bool have = entry < count;
    uint32_t gid = 0;
    float4* rec = a.state;
    Candidate best, neeBest;
    best.prim = neeBest.prim = NO_HIT;
    best.a = best.invDet = best.U = best.V = best.W = 0.0f;
    neeBest.a = neeBest.invDet = neeBest.U = neeBest.V = neeBest.W = 0.0f;
    int next = NEXT_DONE;
    uint32_t word = 0; /* the pixel's state as the trace left it: WF_W_* */
    if (INIT) {
        gid = a.laneFirst + entry;
        uint32_t pixel = a.k.blockStart;
        if (have)
            have = lanePixel(a.k, gid, pixel);
        pathStateInit(ps, pixel, pixel % a.k.width, pixel / a.k.width);
        rec = a.state + (size_t)gid * WF_SLOTS;
        next = NEXT_NEW;
    } else {
        ps.time = 0.0f;
        ps.animCached = -1;
        ps.rayKind = RAY_PATH;
        ps.neeKind = 0;
        ps.o = ps.d = ps.neeO = ps.neeD = mk3(0.0f, 0.0f, 1.0f);
        if (have) {
            gid = queueIn[entry];
            rec = a.state + (size_t)gid * WF_SLOTS;
#pragma unroll
            for (int k = 0; k < SLOT_COUNT; k++)
                ps.base[k * WG] = rec[k];
            const float4 ro = rec[WF_RAY_O], rd = rec[WF_RAY_D], h0 = rec[WF_HIT0], h1 = rec[WF_HIT1];
            word = __float_as_uint(rd.w);
            ps.o = mk3(ro.x, ro.y, ro.z);
            ps.time = ro.w;
            ps.d = mk3(rd.x, rd.y, rd.z);
            if (word & WF_RESUME)
                have = false; /* (walking the ray queue) a suspended walk: the trace has queued it again itself */
            best.prim = __float_as_uint(h0.x);
            best.a = h0.y;
            best.invDet = h0.z;
            best.U = h0.w;
            best.V = h1.x;
            best.W = h1.y;
            ps.neeKind = (int)((word & WF_W_NEE_MASK) >> WF_W_NEE_SHIFT);
            if (ps.neeKind != 0) {
                const float4 n0 = rec[WF_NEE_HIT], no = rec[WF_NEE_O], nd = rec[WF_NEE_D];
                neeBest.prim = __float_as_uint(n0.x);
                neeBest.a = n0.y;
                neeBest.invDet = n0.z;
                neeBest.U = n0.w;
                neeBest.V = h1.z;
                neeBest.W = h1.w;
                ps.neeO = mk3(no.x, no.y, no.z);
                ps.neeD = mk3(nd.x, nd.y, nd.z);
            }
        }
    }
    bool finish = false; /* the pixel's samples are through and a light ray is still out: one more iteration for its answer */
    if (have) {
        LaneCounters lc = LANE_COUNTERS_ZERO;
        if (!INIT) {
            /* the answer of the light ray that travelled beside this path ray, first: the accumulator's additions in the
             * reference's order (wurblpt.hpp:208-218,240-250, then :131-178 of the next component) */
            if (ps.neeKind != 0)
                blockNeeResult<F>(sv, par, tri4, ps, neeBest);
            ps.neeKind = 0;
            if (word & WF_W_PATH)
                next = blockShade<F, false, TriGeomFromScene, PathLds<WG>, true>(sv, par, tri4, ps, best, lc, 0); /* tracePath, one path component (wurblpt.hpp:131-273) */
            else
                next = NEXT_DONE; /* (WF_W_FINISH) nothing but that answer was left */
        }
        if (next == NEXT_NEW) /* the pixel's next sample (wurblpt.hpp:348-360), or nothing more */
            next = blockNew<F>(fa, ps, sv);
        if (next == NEXT_DONE) {
            if (ps.neeKind != 0) {
                finish = true; /* the last path ended behind a hit whose light ray is still to be traced */
            } else {
                /* SensorRGB::finishPixel (sensor_rgb.hpp:82-87) */
                const uint32_t pxy = ps.getW(SLOT_SRDIR);
                const size_t at = (size_t)(pxy >> 16) * a.k.width + (pxy & 0xffffu);
                const Slot acc = ps.get(SLOT_ACC);
                float* out = a.k.frame + 3 * at;
                out[0] = a.k.invSamples * acc.x;
                out[1] = a.k.invSamples * acc.y;
                out[2] = a.k.invSamples * acc.z;
            }
        }
        if (next == NEXT_TRACE || finish) {
#pragma unroll
            for (int k = 0; k < SLOT_COUNT; k++)
                rec[k] = ps.base[k * WG];
            const uint32_t out = (next == NEXT_TRACE ? WF_W_PATH : WF_W_FINISH) | ((uint32_t)ps.neeKind << WF_W_NEE_SHIFT);
            rec[WF_RAY_O] = make_float4(ps.o.x, ps.o.y, ps.o.z, ps.time);
            rec[WF_RAY_D] = make_float4(ps.d.x, ps.d.y, ps.d.z, __uint_as_float(out));
            if (ps.neeKind != 0) {
                rec[WF_NEE_O] = make_float4(ps.neeO.x, ps.neeO.y, ps.neeO.z, 0.0f);
                rec[WF_NEE_D] = make_float4(ps.neeD.x, ps.neeD.y, ps.neeD.z, 0.0f);
            }
        }
    }
    /* ---- the pixels that go on are queued for the next trace: one atomic per workgroup ---- */
    const bool alive = have && (next == NEXT_TRACE || finish);
    const unsigned long long aliveMask = __ballot(alive);
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0)
        ldsWords[wave] = (uint32_t)__popcll(aliveMask);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = ldsWords[0] + ldsWords[1] + ldsWords[2] + ldsWords[3];
        ldsWords[4] = total ? atomicAdd(&nxt->rayCount, total) : 0u;
    }
    __syncthreads();
    if (alive) {
        uint32_t at = ldsWords[4] + rankIn(aliveMask);
        for (uint32_t w = 0; w < wave; w++)
            at += ldsWords[w];
        queueOut[at] = gid;
    }
}

constexpr size_t WF_SHADE_LDS = COLD_BYTES + 32;

/* launch geometry of a wavefront render (wpt_set_wavefront); 0 = default */
struct WfConfig {
    uint32_t groups, chunk, refillIdle, buckets, leafBias, stepBudget, topNodes;
    uint32_t tracePerCu;   /* measurements: workgroups of the trace per compute unit (0 = what fits) */
    uint32_t shadePerKind; /* measurements: one shade launch per kind of material, so that a kernel trace tells them apart */
};
struct WfLaunchers;
/* The host side (wpt_wavefront_host.hip): renders the pixels of `args` (the lane -> pixel mapping of one launch of the
 * single kernel) in wavefront form into args.frame.  Work is enqueued on streams of the calling thread's own that wait
 * for `stream`; the call returns when the frame is complete (the number of iterations is only known as they run).
 * *launches receives the number of kernel launches. */
hipError_t renderWavefront(const KernelArgs& args, const WfLaunchers& kernels, const WfConfig& cfg, hipStream_t stream, uint32_t* launches);

/* The trace kernel is instantiated in ONE translation unit (wpt_k_wf_trace.hip: with and without sphere leaves), so that the
 * shade kernels' units, which differ in their build settings, can never disagree about it. */
void launchWfTrace(bool spheres, const WfArgs& a, dim3 grid, hipStream_t stream);
int wfTraceBlocksPerCu(bool spheres, size_t dynamicLdsBytes);

/* one launcher set per feature set; the shade kernels each in their own translation unit */
struct WfLaunchers {
    bool spheres; /* which instantiation of the trace */
    void (*init)(const WfArgs&, dim3, hipStream_t);
    void (*shade)(const WfArgs&, dim3, hipStream_t);
};
const WfLaunchers& wfBasic();
const WfLaunchers& wfFull();
const WfLaunchers& wfFullRgl();

#define WPT_WF_LAUNCHERS(NAME, FEATURES, SPHERES)                                                                              \
    static void NAME##Init(const WfArgs& a, dim3 grid, hipStream_t s) { hipLaunchKernelGGL((wf_shade<FEATURES, true>), grid, dim3(WG), WF_SHADE_LDS, s, a); } \
    static void NAME##Shade(const WfArgs& a, dim3 grid, hipStream_t s) { hipLaunchKernelGGL((wf_shade<FEATURES, false>), grid, dim3(WG), WF_SHADE_LDS, s, a); } \
    const WfLaunchers& NAME()                                                                                                  \
    {                                                                                                                          \
        static const WfLaunchers l = { SPHERES, NAME##Init, NAME##Shade };                                                     \
        return l;                                                                                                              \
    }

} /* namespace wptk */

#endif
