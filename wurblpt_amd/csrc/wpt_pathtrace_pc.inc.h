/*
 * wpt_pathtrace_pc.inc.h -- path tracing kernel with a ray pool per workgroup
 * ("pc" = every wave is producer and consumer of rays).
 *
 * Why.  In wpt_pathtrace a lane traverses only its own pixel's ray, so whenever that pixel is
 * waiting to be shaded the lane has no traversal work and the traversal loop runs thin
 * (measured: 23-26 of 64 lanes).  Here the two halves of a lane are decoupled:
 *
 *   OWNER half       one pixel per lane: Prng, accumulator, path state, shading
 *                    (blockNew / blockShade / blockNeeEnd of wpt_blocks.h).  When the pixel
 *                    needs a ray traced the lane writes it to its slot in LDS and queues the
 *                    slot number.
 *   TRAVERSAL half   a traversal context per lane (ray, node cursor, best candidate) that
 *                    takes ANY queued ray of the workgroup, walks the BVH for it and writes the
 *                    winning candidate back to that ray's slot.
 *
 * A workgroup of 12 waves pools the rays of 768 pixels, so a wave refills its idle traversal
 * contexts from the pool and keeps its traversal loop full, and it shades only when enough
 * of its own pixels have their result -- waiting costs nothing, the wave traverses meanwhile.
 * Each pixel still performs its own operations in the reference's order (one ray in flight
 * per pixel; the Prng is only touched by its owner lane), so results are unchanged.
 *
 * LDS protocol (one workgroup, LDS is the only shared medium):
 *   slot s (one per owner lane): ray origin + direction in, candidate out (6 floats), flag
 *       0 idle, 1 posted, 2 result ready.
 *   ring[1024]: slot numbers in posting order; qTail reserved by producers with one atomic add
 *       per wave, qHead claimed by consumers with compare-and-swap; an entry is EMPTY until its
 *       producer has written it and is reset by its consumer.  At most 768 rays are posted at
 *       any time (one per owner lane), so the ring cannot overflow.
 *   Data is written before the flag / ring entry that publishes it, with workgroup-scope
 *   release/acquire fences; LDS operations of one wave complete in order.
 * Every wait is bounded: a wave that spins too long raises `abortFlag`, every loop checks it,
 * the kernel drains and the launch reports failure (wpt_scene_check) instead of hanging the GPU.
 */
#ifndef WPT_PATHTRACE_PC_INC_H
#define WPT_PATHTRACE_PC_INC_H

#include <hip/hip_runtime.h>

#include "../../include/wurblpt_hip.h"
#include "wpt_blocks.h"
#include "wpt_pathtrace.inc.h"

namespace wptk {

constexpr int PC_WAVES = 12;
constexpr int PC_WG = PC_WAVES * 64; /* 768 threads = 768 pixels per workgroup */
constexpr int PC_SLOTS = PC_WG;
constexpr int PC_RING = 1024;
constexpr uint32_t PC_EMPTY = 0xffffffffu;
constexpr uint32_t PC_SPIN_LIMIT = 1u << 24;

struct PcShared {
    uint32_t ring[PC_RING];
    float slot[6][PC_SLOTS];
    uint32_t slotFlag[PC_SLOTS];
    uint32_t qHead, qTail;
    uint32_t abortFlag;
};

WPT_D void pcRelease() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
WPT_D void pcAcquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
WPT_D uint32_t pcLoad(uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WPT_D void pcStore(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template<uint32_t F, bool LDSSCENE>
__global__ __launch_bounds__(PC_WG, 3) void wpt_pathtrace_pc(const KernelArgs args)
{
    extern __shared__ float4 ldsScene[];
    __shared__ PcShared sh;

    const SceneView& sv = args.sv;
    const wpt_params& par = args.par;
    const uint32_t nodeCount = sv.nodeCount;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;

    for (uint32_t i = tid; i < PC_RING; i += PC_WG)
        sh.ring[i] = PC_EMPTY;
    sh.slotFlag[tid] = 0;
    if (tid == 0) {
        sh.qHead = 0;
        sh.qTail = 0;
        sh.abortFlag = 0;
    }
    if (LDSSCENE) {
        const uint32_t n4 = 2 * nodeCount, t4 = 3 * sv.triCount;
        for (uint32_t i = tid; i < n4; i += PC_WG)
            ldsScene[i] = sv.nodes[i];
        for (uint32_t i = tid; i < t4; i += PC_WG)
            ldsScene[n4 + i] = sv.triGeom[i];
    }
    __syncthreads();
    auto node4 = [&](uint32_t i) -> float4 {
        if constexpr (LDSSCENE)
            return ldsScene[i];
        else
            return sv.nodes[i];
    };
    auto tri4 = [&](uint32_t i) -> float4 {
        if constexpr (LDSSCENE)
            return ldsScene[2 * nodeCount + i];
        else
            return sv.triGeom[i];
    };

    /* ---- owner half ---- */
    const uint32_t slot = tid;
    const uint32_t gid = blockIdx.x * PC_SLOTS + slot;
    const bool inBlock = gid < args.blockSize;
    uint32_t pixel;
    if (args.tiled) {
        const uint32_t tilesPerRow = args.width >> 3;
        const uint32_t tile = gid >> 6;
        const uint32_t tx = tile % tilesPerRow, ty = tile / tilesPerRow;
        pixel = args.blockStart + ((ty << 3) + (lane >> 3)) * args.width + (tx << 3) + (lane & 7u);
    } else {
        pixel = args.blockStart + gid;
    }
    if (!inBlock)
        pixel = args.blockStart;
    FrameArgs fa;
    fa.cam = args.cam;
    fa.par = args.par;
    fa.width = args.width;
    fa.height = args.height;
    fa.samplesSqrt = args.samplesSqrt;
    PathState ps;
    pathStateInit(ps, pixel, args.width);
    LaneCounters lc = { 0, 0, 0, 0, 0, { 0, 0, 0, 0, 0, 0, 0, 0 } };
    enum { O_NEW = 0, O_WAIT = 1, O_READY = 2, O_DONE = 3 };
    int ost = inBlock ? O_NEW : O_DONE;

    /* ---- traversal half ---- */
    enum { T_IDLE = 0, T_NODE = 1, T_LEAF = 2 };
    int tst = T_IDLE;
    uint32_t tslot = 0;
    f3 org = mk3(0.0f, 0.0f, 0.0f);
    RayAux aux = rayAux(mk3(0.0f, 0.0f, 1.0f));
    uint32_t node = 0, leafPrim = 0;
    float amax = k_maxval;
    Candidate best;
    best.prim = NO_HIT;
    best.a = best.invDet = best.U = best.V = best.W = 0.0f;

    uint32_t idleSpins = 0;
    int patience = 0;
    /* optional wave-level statistics (args.schedStats): same layout as the single-role kernel */
    const bool stats = args.schedStats != nullptr;
    unsigned long long sched[11] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };

    for (;;) {
        if (pcLoad(&sh.abortFlag))
            break;
        /* results that have arrived for this wave's pixels */
        if (ost == O_WAIT && pcLoad(&sh.slotFlag[slot]) == 2u)
            ost = O_READY;
        const int cShade = __popcll(__ballot(ost == O_READY && ps.rayKind == RAY_PATH));
        const int cNee = __popcll(__ballot(ost == O_READY && ps.rayKind != RAY_PATH));
        const int cNew = __popcll(__ballot(ost == O_NEW));
        const int cWait = __popcll(__ballot(ost == O_WAIT));
        const int nBusy = __popcll(__ballot(tst != T_IDLE));
        const bool queued = pcLoad(&sh.qTail) != pcLoad(&sh.qHead);
        if ((cShade | cNee | cNew | cWait | nBusy) == 0) {
            /* this wave's pixels are finished; keep serving the pool until it is empty.  Other
             * waves' pixels post rays only after their own shading, which needs no help. */
            if (!queued)
                break;
        }
        /* Shade when a block is well filled; or in a hurry: nothing to traverse (no busy
         * context, nothing queued) or patience ran out. */
        const int ownMin = (int)args.heavyMin < 1 ? 1 : (int)args.heavyMin;
        const bool hurry = (nBusy == 0 && !queued) || patience > (int)args.patience;
        int pick = -1; /* 0 shade, 1 nee-end, 2 new */
        if (cShade >= ownMin || (hurry && cShade > 0 && cShade >= cNee && cShade >= cNew))
            pick = 0;
        else if (cNee >= ownMin || (hurry && cNee > 0 && cNee >= cNew))
            pick = 1;
        else if (cNew >= ownMin || (hurry && cNew > 0))
            pick = 2;

        if (pick >= 0) {
            patience = 0;
            idleSpins = 0;
            if (stats) {
                sched[5 + 2 * pick]++;
                sched[6 + 2 * pick] += pick == 0 ? cShade : pick == 1 ? cNee : cNew;
            }
            int next = -1;
            const bool mine = ost == O_READY && ((pick == 0) == (ps.rayKind == RAY_PATH)) && pick != 2;
            if (mine) {
                /* the candidate the traversal left in this pixel's slot */
                pcAcquire();
                Candidate res;
                res.prim = __float_as_uint(sh.slot[0][slot]);
                res.a = sh.slot[1][slot];
                res.invDet = sh.slot[2][slot];
                res.U = sh.slot[3][slot];
                res.V = sh.slot[4][slot];
                res.W = sh.slot[5][slot];
                pcStore(&sh.slotFlag[slot], 0u);
                if (pick == 0)
                    next = blockShade<F, false>(sv, par, tri4, ps, res, lc);
                else
                    next = blockNeeEnd<F>(sv, par, ps, res);
            } else if (pick == 2 && ost == O_NEW) {
                next = blockNew<F>(fa, ps);
            }
            const bool posting = next == NEXT_TRACE;
            if (next == NEXT_TRACE)
                ost = O_WAIT;
            else if (next == NEXT_NEW)
                ost = O_NEW;
            else if (next == NEXT_DONE)
                ost = O_DONE;
            /* post the rays: slot data, then the ring entry that publishes it */
            const unsigned long long mask = __ballot(posting);
            const uint32_t n = __popcll(mask);
            if (n > 0) {
                if (posting) {
                    sh.slot[0][slot] = ps.ray.o.x;
                    sh.slot[1][slot] = ps.ray.o.y;
                    sh.slot[2][slot] = ps.ray.o.z;
                    sh.slot[3][slot] = ps.ray.d.x;
                    sh.slot[4][slot] = ps.ray.d.y;
                    sh.slot[5][slot] = ps.ray.d.z;
                    pcStore(&sh.slotFlag[slot], 1u);
                }
                pcRelease();
                uint32_t base = 0;
                if (lane == 0)
                    base = __hip_atomic_fetch_add(&sh.qTail, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                base = __builtin_amdgcn_readfirstlane(base);
                if (posting) {
                    const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
                    uint32_t* entry = &sh.ring[(base + rank) & (PC_RING - 1)];
                    uint32_t spins = 0;
                    while (pcLoad(entry) != PC_EMPTY) { /* its previous consumer has not reset it yet */
                        if (++spins > PC_SPIN_LIMIT || pcLoad(&sh.abortFlag)) {
                            pcStore(&sh.abortFlag, 1u);
                            break;
                        }
                    }
                    pcStore(entry, slot);
                }
            }
            continue;
        }

        /* ---- traversal work ---- */
        if (nBusy <= 48 && queued) {
            /* refill idle contexts: claim up to 64 - nBusy queued rays */
            const uint32_t want = 64u - (uint32_t)nBusy;
            uint32_t base = 0, take = 0;
            if (lane == 0) {
                uint32_t h = pcLoad(&sh.qHead);
                for (int tries = 0; tries < 64; tries++) {
                    const uint32_t t = pcLoad(&sh.qTail);
                    const uint32_t avail = t - h;
                    take = avail < want ? avail : want;
                    if (take == 0)
                        break;
                    if (__hip_atomic_compare_exchange_strong(&sh.qHead, &h, h + take, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                        base = h;
                        break;
                    }
                    take = 0; /* h was refreshed by the failed exchange */
                }
            }
            base = __builtin_amdgcn_readfirstlane(base);
            take = __builtin_amdgcn_readfirstlane(take);
            if (take > 0) {
                const unsigned long long idleMask = __ballot(tst == T_IDLE);
                const uint32_t rank = __popcll(idleMask & ((1ull << lane) - 1ull));
                if (tst == T_IDLE && rank < take) {
                    uint32_t* entry = &sh.ring[(base + rank) & (PC_RING - 1)];
                    uint32_t s = pcLoad(entry);
                    uint32_t spins = 0;
                    while (s == PC_EMPTY) { /* reserved by its producer but not written yet */
                        if (++spins > PC_SPIN_LIMIT || pcLoad(&sh.abortFlag)) {
                            pcStore(&sh.abortFlag, 1u);
                            break;
                        }
                        s = pcLoad(entry);
                    }
                    if (s != PC_EMPTY) {
                        pcStore(entry, PC_EMPTY);
                        pcAcquire();
                        tslot = s;
                        org = mk3(sh.slot[0][s], sh.slot[1][s], sh.slot[2][s]);
                        const f3 dir = mk3(sh.slot[3][s], sh.slot[4][s], sh.slot[5][s]);
                        aux = rayAux(dir);
                        node = 0;
                        amax = k_maxval;
                        best.prim = NO_HIT;
                        tst = T_NODE;
                    }
                }
            }
        }
        const int nWalk = __popcll(__ballot(tst != T_IDLE));
        if (nWalk == 0) {
            /* nothing to traverse, nothing worth shading yet: wait for other waves */
            __builtin_amdgcn_s_sleep(2);
            patience++;
            if (++idleSpins > PC_SPIN_LIMIT) {
                pcStore(&sh.abortFlag, 1u);
                break;
            }
            continue;
        }
        idleSpins = 0;
        patience++;
        if (stats)
            sched[0]++;
        /* walk: NODE steps and LEAF tests; leave to refill (or to look at the owner half) when
         * a quarter of the entering contexts has finished, at the latest after 64 steps */
        const int leaveBelow = nWalk - ((nWalk + 3) >> 2);
        for (int steps = 0; steps < 64; steps++) {
            const int nNode = __popcll(__ballot(tst == T_NODE));
            const int nLeaf = __popcll(__ballot(tst == T_LEAF));
            if (nNode + nLeaf <= leaveBelow)
                break;
            bool finished = false;
            if (nLeaf * (int)args.leafBias >= nNode * 8 && nLeaf > 0) {
                if (stats) {
                    sched[3]++;
                    sched[4] += nLeaf;
                }
                if (tst == T_LEAF) {
                    /* HitableTriangle::hit, candidate part (hitable_triangle.hpp:189-271) */
                    const float4 g0 = tri4(3 * leafPrim), g1 = tri4(3 * leafPrim + 1), g2 = tri4(3 * leafPrim + 2);
                    Candidate c;
                    if (triangleTest(mk3(g0.x, g0.y, g0.z), mk3(g1.x, g1.y, g1.z), mk3(g2.x, g2.y, g2.z), org, aux,
                                par.min_hit_distance, amax, c)) {
                        c.prim = leafPrim;
                        best = c;
                        amax = c.a;
                    }
                    node = node + 1;
                    tst = T_NODE;
                    finished = node >= nodeCount;
                }
            } else {
                if (stats) {
                    sched[1]++;
                    sched[2] += nNode;
                }
                if (tst == T_NODE) {
                    /* AABB::mayHit + the stackless form of BVH::hit's walk */
                    const float4 n0 = node4(2 * node);
                    const float4 n1 = node4(2 * node + 1);
                    const uint32_t skip = __float_as_uint(n1.z);
                    const uint32_t prim = __float_as_uint(n1.w);
                    const bool hit = boxTest(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), org, aux.inv, par.min_hit_distance, amax);
                    const bool toLeaf = hit && prim < NODE_EMPTY;
                    const uint32_t nextNode = (hit && prim == NODE_INNER) ? node + 1 : skip;
                    leafPrim = toLeaf ? prim : leafPrim;
                    node = toLeaf ? node : nextNode;
                    tst = toLeaf ? (int)T_LEAF : (int)T_NODE;
                    finished = !toLeaf && node >= nodeCount;
                }
            }
            if (finished) {
                /* hand the winning candidate back to the ray's owner */
                sh.slot[0][tslot] = __uint_as_float(best.prim);
                sh.slot[1][tslot] = best.a;
                sh.slot[2][tslot] = best.invDet;
                sh.slot[3][tslot] = best.U;
                sh.slot[4][tslot] = best.V;
                sh.slot[5][tslot] = best.W;
                pcRelease();
                pcStore(&sh.slotFlag[tslot], 2u);
                tst = T_IDLE;
            }
        }
    }
    if (inBlock && !pcLoad(&sh.abortFlag)) {
        /* SensorRGB::finishPixel (sensor_rgb.hpp:82-87) */
        const float invSamples = 1.0f / (float)(args.samplesSqrt * args.samplesSqrt);
        float* out = args.frame + 3 * (size_t)pixel;
        out[0] = invSamples * ps.acc0;
        out[1] = invSamples * ps.acc1;
        out[2] = invSamples * ps.acc2;
    }
    if (stats && lane == 0) {
        for (int i = 0; i < 11; i++)
            atomicAdd(args.schedStats + i, sched[i]);
    }
    /* a launch that had to abort must not look like a finished frame */
    if (pcLoad(&sh.abortFlag) && args.status && tid == 0)
        atomicExch(args.status, 1u);
}

void launchPcBasicLds(const KernelArgs& args, dim3 grid, size_t ldsBytes, hipStream_t stream);
void launchPcBasic(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchPcFull(const KernelArgs& args, dim3 grid, hipStream_t stream);

} /* namespace wptk */

#endif
