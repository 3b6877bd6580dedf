/*
 * wpt_pathtrace_wf.inc.h -- path tracing kernel with the pixel states of a workgroup in LDS
 * ("wf": a wavefront path tracer inside one compute unit).
 *
 * Why.  In wpt_pathtrace a lane owns a pixel, so a wave's 64 lanes are spread over all phases
 * of a path and every piece of code runs with about a third of the lanes (measured: 21-25
 * of 64).  Here a pixel's state (Prng, accumulator, path state: 45 words) lives in LDS, not in a
 * lane, and waves pick up whichever pixels are ready for the code they are about to run:
 *
 *   heavy batch   64 pixels whose ray has come back (or that need a new sample) are taken
 *                 from the HEAVY queue, their state is loaded into registers, the block
 *                 functions of wpt_blocks.h run for all 64 lanes, the state goes back to LDS and
 *                 the pixels whose next ray is ready go to the TRAV queue.
 *   traversal     the first `travWaves` waves keep one traversal context per lane (ray, node
 *                 cursor, best candidate).  Idle contexts are refilled from the TRAV queue, a
 *                 quarter of the wave at a time; a finished context writes the winning candidate
 *                 to its pixel's state and sends the pixel to the HEAVY queue.
 *
 * One workgroup of 16 waves per compute unit holds 768 pixel slots (135 KiB of the 160 KiB LDS);
 * a slot whose pixel has finished all samples takes the next pixel of the block from a global
 * counter, so the grid is persistent and balances itself.  There are fewer traversal contexts
 * than pixels on purpose: the queues stay filled and the waves stay full.
 * Every pixel still performs its own operations in the reference's order (one ray in flight per
 * pixel, its Prng is touched only by the batch that holds it), so results do not change.
 *
 * LDS protocol: two rings of slot numbers (TRAV, HEAVY).  Producers reserve ring entries with one
 * atomic add per wave; consumers claim with compare-and-swap on the head; an entry is EMPTY until
 * its producer has written it and is reset by its consumer.  A slot is in at most one ring, so a
 * ring of 1024 cannot overflow.  State is written before the ring entry that publishes it
 * (workgroup-scope release / acquire fences).  Every wait is bounded: a wave that waits too long
 * raises `abortFlag`, all loops watch it, the launch drains and reports failure
 * (wpt_scene_check) instead of hanging the GPU.
 */
#ifndef WPT_PATHTRACE_WF_INC_H
#define WPT_PATHTRACE_WF_INC_H

#include <hip/hip_runtime.h>

#include "../../include/wurblpt_hip.h"
#include "wpt_blocks.h"
#include "wpt_pathtrace.inc.h"

namespace wptk {

constexpr int WF_WAVES = 16;
constexpr int WF_WG = WF_WAVES * 64;
constexpr int WF_SLOTS = 768;
constexpr int WF_RING = 1024;
constexpr int WF_FIELDS = 45;
constexpr uint32_t WF_EMPTY = 0xffffffffu;
constexpr uint32_t WF_SPIN_LIMIT = 1u << 24;
constexpr uint32_t WF_LDS_SCENE_MAX_BYTES = 8 * 1024;
constexpr uint32_t WF_META_INIT = 0x80000000u;

/* words of LDS behind the scene copy */
constexpr uint32_t WF_OFF_STATE = 0;
constexpr uint32_t WF_OFF_RING_T = WF_OFF_STATE + WF_FIELDS * WF_SLOTS;
constexpr uint32_t WF_OFF_RING_H = WF_OFF_RING_T + WF_RING;
constexpr uint32_t WF_OFF_CTL = WF_OFF_RING_H + WF_RING;
constexpr uint32_t WF_WORDS = WF_OFF_CTL + 16;
enum { WF_T_HEAD = 0, WF_T_TAIL = 1, WF_H_HEAD = 2, WF_H_TAIL = 3, WF_DONE_SLOTS = 4, WF_ABORT = 5, WF_ACTIVE = 6 };

/* state fields (one word each, stored as field-major arrays over the slots) */
enum {
    WF_F_PRNG = 0, WF_F_ACC = 4, WF_F_PIXEL = 7, WF_F_SAMPLE = 8, WF_F_META = 9, WF_F_ORG = 10, WF_F_DIR = 13,
    WF_F_RI = 16, WF_F_ATT = 20, WF_F_OPL = 24, WF_F_NEXTATT = 27, WF_F_NEEFACTOR = 31, WF_F_SRDIR = 35,
    WF_F_CHOSEN = 38, WF_F_BEST = 39
};

inline size_t wfLdsBytes(uint32_t sceneBytes) { return size_t(sceneBytes) + size_t(WF_WORDS) * 4; }

WPT_D void wfRelease() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
WPT_D void wfAcquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
WPT_D uint32_t wfLoad(uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WPT_D void wfStore(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

/* entries in a ring: the head is read first (it never passes the tail) */
WPT_D uint32_t wfCount(uint32_t* head, uint32_t* tail)
{
    const uint32_t h = __hip_atomic_load(head, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t t = __hip_atomic_load(tail, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t n = t - h;
    return (int32_t)n < 0 ? 0u : n;
}

/* Appends the slots of the lanes with `posting` to a ring: one reservation per wave, then
 * every lane waits (bounded) until its entry has been reset by its previous consumer. */
WPT_D void wfPush(uint32_t* ring, uint32_t* tail, uint32_t* abortFlag, bool posting, uint32_t slot, uint32_t lane, uint32_t ringMask = WF_RING - 1)
{
    const unsigned long long mask = __ballot(posting);
    const uint32_t n = __popcll(mask);
    if (n == 0)
        return;
    uint32_t base = 0;
    if (lane == 0)
        base = __hip_atomic_fetch_add(tail, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    base = __builtin_amdgcn_readfirstlane(base);
    if (posting) {
        const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
        uint32_t* entry = &ring[(base + rank) & ringMask];
        uint32_t spins = 0;
        while (wfLoad(entry) != WF_EMPTY) {
            if (++spins > WF_SPIN_LIMIT || wfLoad(abortFlag)) {
                wfStore(abortFlag, 1u);
                break;
            }
        }
        wfStore(entry, slot);
    }
}

/* Claims up to `want` entries of a ring for the lanes with `taker` (in lane order).  Returns the
 * slot for a lane that got one, WF_EMPTY otherwise. */
WPT_D uint32_t wfPop(uint32_t* ring, uint32_t* head, uint32_t* tail, uint32_t* abortFlag, bool taker, uint32_t want, uint32_t lane, uint32_t ringMask = WF_RING - 1)
{
    uint32_t base = 0, take = 0;
    if (lane == 0) {
        uint32_t h = __hip_atomic_load(head, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (int tries = 0; tries < 64; tries++) {
            const uint32_t t = __hip_atomic_load(tail, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t avail = (int32_t)(t - h) < 0 ? 0u : t - h;
            take = avail < want ? avail : want;
            if (take == 0)
                break;
            if (__hip_atomic_compare_exchange_strong(head, &h, h + take, __ATOMIC_ACQUIRE, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                base = h;
                break;
            }
            take = 0; /* h was refreshed by the failed exchange */
        }
    }
    base = __builtin_amdgcn_readfirstlane(base);
    take = __builtin_amdgcn_readfirstlane(take);
    uint32_t got = WF_EMPTY;
    if (take > 0) {
        const unsigned long long takers = __ballot(taker);
        const uint32_t rank = __popcll(takers & ((1ull << lane) - 1ull));
        if (taker && rank < take) {
            uint32_t* entry = &ring[(base + rank) & ringMask];
            uint32_t s = wfLoad(entry);
            uint32_t spins = 0;
            while (s == WF_EMPTY) { /* reserved by its producer but not written yet */
                if (++spins > WF_SPIN_LIMIT || wfLoad(abortFlag)) {
                    wfStore(abortFlag, 1u);
                    break;
                }
                s = wfLoad(entry);
            }
            if (s != WF_EMPTY) {
                wfStore(entry, WF_EMPTY);
                got = s;
            }
        }
    }
    return got;
}

/* GSTATE: the pixel states live in global memory (one 192-byte record per slot, read and written
 * with 16-byte accesses; args.wfState, args.wfSlots per workgroup) instead of LDS, so that a
 * workgroup can hold several times more pixels than traversal contexts */
constexpr int WF_GRING = 4096;      /* ring size with GSTATE = most slots a workgroup may have */
constexpr int WF_GRECORD4 = 12;     /* float4 per slot record */

template<uint32_t F, bool LDSSCENE, bool GSTATE = false>
__global__ __launch_bounds__(WF_WG) void wpt_pathtrace_wf(const KernelArgs args)
{
    constexpr uint32_t RING = GSTATE ? (uint32_t)WF_GRING : (uint32_t)WF_RING;
    extern __shared__ float4 ldsRaw[];

    const SceneView& sv = args.sv;
    const wpt_params& par = args.par;
    const uint32_t nodeCount = sv.nodeCount;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    /* LDS: [scene copy (LDSSCENE)] [state] [rings] [control words] */
    const uint32_t scene4 = LDSSCENE ? 2 * nodeCount + 3 * sv.triCount : 0u;
    float4* ldsScene = ldsRaw;
    uint32_t* words = reinterpret_cast<uint32_t*>(ldsRaw + scene4);
    uint32_t* state = words + WF_OFF_STATE;                      /* !GSTATE only */
    uint32_t* ringT = words + (GSTATE ? 0u : WF_OFF_RING_T);
    uint32_t* ringH = ringT + RING;
    uint32_t* ctl = ringH + RING;
    uint32_t* abortFlag = ctl + WF_ABORT;
    const uint32_t slots = GSTATE ? (args.wfSlots < 64u ? 64u : (args.wfSlots > RING ? RING : args.wfSlots)) : (uint32_t)WF_SLOTS;
    float4* grec = GSTATE ? args.wfState + (size_t)blockIdx.x * slots * WF_GRECORD4 : nullptr;

    for (uint32_t i = tid; i < RING; i += WF_WG) {
        ringT[i] = WF_EMPTY;
        ringH[i] = i < slots ? i : WF_EMPTY; /* every slot starts in the HEAVY queue, asking for a pixel */
    }
    for (uint32_t i = tid; i < slots; i += WF_WG) {
        if constexpr (GSTATE)
            reinterpret_cast<uint32_t*>(grec + (size_t)i * WF_GRECORD4 + 1)[2] = WF_META_INIT;
        else
            state[WF_F_META * WF_SLOTS + i] = WF_META_INIT;
    }
    if (tid < 16)
        ctl[tid] = tid == WF_H_TAIL ? slots : 0u;
    if (GSTATE)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (LDSSCENE) {
        const uint32_t n4 = 2 * nodeCount, t4 = 3 * sv.triCount;
        for (uint32_t i = tid; i < n4; i += WF_WG)
            ldsScene[i] = sv.nodes[i];
        for (uint32_t i = tid; i < t4; i += WF_WG)
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
    auto fld = [&](uint32_t f, uint32_t s) -> uint32_t& { return state[f * WF_SLOTS + s]; };
    auto fldF = [&](uint32_t f, uint32_t s) -> float& { return reinterpret_cast<float*>(state)[f * WF_SLOTS + s]; };

    FrameArgs fa;
    fa.cam = args.cam;
    fa.par = args.par;
    fa.width = args.width;
    fa.height = args.height;
    fa.samplesSqrt = args.samplesSqrt;
    const float invSamples = 1.0f / (float)(args.samplesSqrt * args.samplesSqrt);
    const uint32_t travWaves = args.travWaves < 1 ? 1u : (args.travWaves > (uint32_t)WF_WAVES ? (uint32_t)WF_WAVES : args.travWaves);
    const bool travWave = wave < travWaves;
    /* waves behind the traversal waves run heavy batches only; the ones not asked for leave */
    const uint32_t heavyWaves = args.heavyWaves < 1 ? 1u : args.heavyWaves;
    if (wave >= travWaves + heavyWaves)
        return;
    const uint32_t capacity = travWaves * 64u;
    LaneCounters lc = { 0, 0, 0, 0, 0, { 0, 0, 0, 0, 0, 0, 0, 0 } };

    /* ---- traversal contexts (waves below travWaves) ---- */
    enum { T_IDLE = 0, T_NODE = 1, T_LEAF = 2, T_FIN = 3 };
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
    uint32_t patience = 0;
    const bool stats = args.schedStats != nullptr;
    unsigned long long sched[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };

    for (;;) {
        if (wfLoad(abortFlag))
            break;
        const int nAct = __popcll(__ballot(tst == T_NODE || tst == T_LEAF));
        const int nFin = __popcll(__ballot(tst == T_FIN));
        /* finished rays go back to their pixels once a quarter of the wave has them, or when
         * the wave has nothing else to do */
        if (nFin > 0 && (nFin >= 16 || nAct == 0)) {
            const bool fin = tst == T_FIN;
            if (fin) {
                if constexpr (GSTATE) {
                    /* record: q2 = prim a invDet U, q3.xy = V W */
                    float4* rp = grec + (size_t)tslot * WF_GRECORD4;
                    rp[2] = make_float4(__uint_as_float(best.prim), best.a, best.invDet, best.U);
                    *reinterpret_cast<float2*>(rp + 3) = make_float2(best.V, best.W);
                } else {
                    fld(WF_F_BEST + 0, tslot) = best.prim;
                    fldF(WF_F_BEST + 1, tslot) = best.a;
                    fldF(WF_F_BEST + 2, tslot) = best.invDet;
                    fldF(WF_F_BEST + 3, tslot) = best.U;
                    fldF(WF_F_BEST + 4, tslot) = best.V;
                    fldF(WF_F_BEST + 5, tslot) = best.W;
                }
                tst = T_IDLE;
            }
            wfRelease();
            wfPush(ringH, ctl + WF_H_TAIL, abortFlag, fin, tslot, lane, RING - 1);
            if (lane == 0)
                __hip_atomic_fetch_sub(ctl + WF_ACTIVE, (uint32_t)nFin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const uint32_t hv = wfCount(ctl + WF_H_HEAD, ctl + WF_H_TAIL);
        const uint32_t tr = wfCount(ctl + WF_T_HEAD, ctl + WF_T_TAIL);

        /* ---- heavy batch: a heavy wave takes 64 ready pixels; fewer when the traversal side is
         * running dry (rays in flight + queued below 3/4 of the contexts), so that it is fed ---- */
        bool heavy = false;
        if (!travWave) {
            const uint32_t inFlight = wfLoad(ctl + WF_ACTIVE) + tr;
            if (hv >= 64u)
                heavy = true;
            else if (hv > 0u && inFlight * 4u < capacity * 3u && (hv >= 16u || patience > args.patience))
                heavy = true;
        }
        if (heavy) {
            const uint32_t s = wfPop(ringH, ctl + WF_H_HEAD, ctl + WF_H_TAIL, abortFlag, true, 64u, lane, RING - 1);
            const bool mine = s != WF_EMPTY;
            if (__ballot(mine) == 0)
                continue; /* another wave was faster */
            patience = 0;
            idleSpins = 0;
            wfAcquire();
            PathState ps;
            Candidate res;
            uint32_t meta = WF_META_INIT, pixel = 0;
            ps.prng.s0 = ps.prng.s1 = ps.prng.s2 = ps.prng.s3 = 0;
            ps.acc0 = ps.acc1 = ps.acc2 = 0.0f;
            ps.px = ps.py = ps.sampleIndex = ps.pathComponent = 0;
            ps.rayKind = RAY_PATH;
            ps.ray.o = ps.ray.d = ps.opl = ps.srDir = mk3(0.0f, 0.0f, 0.0f);
            ps.ray.ri = ps.att = ps.nextAtt = ps.neeFactor = mk4(0.0f, 0.0f, 0.0f, 0.0f);
            ps.chosenPrim = NO_HIT;
            res.prim = NO_HIT;
            res.a = res.invDet = res.U = res.V = res.W = 0.0f;
            if (mine) {
                if constexpr (GSTATE) {
                    /* record layout (12 x float4): q0 org.xyz dir.x | q1 dir.yz meta pixel | q2 best.prim a invDet U |
                     * q3 best.V W sample chosen | q4 prng | q5 acc | q6 ri | q7 att | q8 opl | q9 nextAtt | q10 neeFactor | q11 srDir */
                    const float4* rp = grec + (size_t)s * WF_GRECORD4;
                    const float4 q1 = rp[1];
                    meta = __float_as_uint(q1.z);
                    if (!(meta & WF_META_INIT)) {
                        const float4 q0 = rp[0], q2 = rp[2], q3 = rp[3], q4 = rp[4], q5 = rp[5], q6 = rp[6], q7 = rp[7], q8 = rp[8],
                                     q9 = rp[9], q10 = rp[10], q11 = rp[11];
                        ps.ray.o = mk3(q0.x, q0.y, q0.z);
                        ps.ray.d = mk3(q0.w, q1.x, q1.y);
                        pixel = __float_as_uint(q1.w);
                        ps.px = pixel % args.width;
                        ps.py = pixel / args.width;
                        ps.pathComponent = meta & 0xffffu;
                        ps.rayKind = (int)((meta >> 16) & 3u);
                        res.prim = __float_as_uint(q2.x);
                        res.a = q2.y;
                        res.invDet = q2.z;
                        res.U = q2.w;
                        res.V = q3.x;
                        res.W = q3.y;
                        ps.sampleIndex = __float_as_uint(q3.z);
                        ps.chosenPrim = __float_as_uint(q3.w);
                        ps.prng.s0 = __float_as_uint(q4.x);
                        ps.prng.s1 = __float_as_uint(q4.y);
                        ps.prng.s2 = __float_as_uint(q4.z);
                        ps.prng.s3 = __float_as_uint(q4.w);
                        ps.acc0 = q5.x;
                        ps.acc1 = q5.y;
                        ps.acc2 = q5.z;
                        ps.ray.ri = mk4(q6.x, q6.y, q6.z, q6.w);
                        ps.att = mk4(q7.x, q7.y, q7.z, q7.w);
                        ps.opl = mk3(q8.x, q8.y, q8.z);
                        ps.nextAtt = mk4(q9.x, q9.y, q9.z, q9.w);
                        ps.neeFactor = mk4(q10.x, q10.y, q10.z, q10.w);
                        ps.srDir = mk3(q11.x, q11.y, q11.z);
                    }
                } else {
                    meta = fld(WF_F_META, s);
                    if (!(meta & WF_META_INIT)) {
                        ps.prng.s0 = fld(WF_F_PRNG + 0, s);
                        ps.prng.s1 = fld(WF_F_PRNG + 1, s);
                        ps.prng.s2 = fld(WF_F_PRNG + 2, s);
                        ps.prng.s3 = fld(WF_F_PRNG + 3, s);
                        ps.acc0 = fldF(WF_F_ACC + 0, s);
                        ps.acc1 = fldF(WF_F_ACC + 1, s);
                        ps.acc2 = fldF(WF_F_ACC + 2, s);
                        pixel = fld(WF_F_PIXEL, s);
                        ps.px = pixel % args.width;
                        ps.py = pixel / args.width;
                        ps.sampleIndex = fld(WF_F_SAMPLE, s);
                        ps.pathComponent = meta & 0xffffu;
                        ps.rayKind = (int)((meta >> 16) & 3u);
                        ps.ray.o = mk3(fldF(WF_F_ORG + 0, s), fldF(WF_F_ORG + 1, s), fldF(WF_F_ORG + 2, s));
                        ps.ray.d = mk3(fldF(WF_F_DIR + 0, s), fldF(WF_F_DIR + 1, s), fldF(WF_F_DIR + 2, s));
                        ps.ray.ri = mk4(fldF(WF_F_RI + 0, s), fldF(WF_F_RI + 1, s), fldF(WF_F_RI + 2, s), fldF(WF_F_RI + 3, s));
                        ps.att = mk4(fldF(WF_F_ATT + 0, s), fldF(WF_F_ATT + 1, s), fldF(WF_F_ATT + 2, s), fldF(WF_F_ATT + 3, s));
                        ps.opl = mk3(fldF(WF_F_OPL + 0, s), fldF(WF_F_OPL + 1, s), fldF(WF_F_OPL + 2, s));
                        ps.nextAtt = mk4(fldF(WF_F_NEXTATT + 0, s), fldF(WF_F_NEXTATT + 1, s), fldF(WF_F_NEXTATT + 2, s), fldF(WF_F_NEXTATT + 3, s));
                        ps.neeFactor = mk4(fldF(WF_F_NEEFACTOR + 0, s), fldF(WF_F_NEEFACTOR + 1, s), fldF(WF_F_NEEFACTOR + 2, s), fldF(WF_F_NEEFACTOR + 3, s));
                        ps.srDir = mk3(fldF(WF_F_SRDIR + 0, s), fldF(WF_F_SRDIR + 1, s), fldF(WF_F_SRDIR + 2, s));
                        ps.chosenPrim = fld(WF_F_CHOSEN, s);
                        res.prim = fld(WF_F_BEST + 0, s);
                        res.a = fldF(WF_F_BEST + 1, s);
                        res.invDet = fldF(WF_F_BEST + 2, s);
                        res.U = fldF(WF_F_BEST + 3, s);
                        res.V = fldF(WF_F_BEST + 4, s);
                        res.W = fldF(WF_F_BEST + 5, s);
                    }
                }
            }
            const bool init = mine && (meta & WF_META_INIT) != 0;
            const bool live = mine && !init;
            if (stats) {
                sched[5]++;
                sched[6] += __popcll(__ballot(live && ps.rayKind == RAY_PATH));
                sched[7]++;
                sched[8] += __popcll(__ballot(live && ps.rayKind != RAY_PATH));
            }
            int next = NEXT_DONE;
            const bool pathRay = ps.rayKind == RAY_PATH; /* the kind of ray that has come back */
            if (live && pathRay)
                next = blockShade<F, false>(sv, par, tri4, ps, res, lc);
            if (live && !pathRay)
                next = blockNeeEnd<F>(sv, par, ps, res);
            if (live && next == NEXT_NEW)
                next = blockNew<F>(fa, ps);
            /* pixels that have finished all their samples (and slots that never had one) */
            const bool needPixel = mine && next == NEXT_DONE;
            const unsigned long long needMask = __ballot(needPixel);
            bool dead = false;
            if (needMask != 0) {
                if (needPixel && live) {
                    /* SensorRGB::finishPixel (sensor_rgb.hpp:82-87) */
                    float* out = args.frame + 3 * (size_t)pixel;
                    out[0] = invSamples * ps.acc0;
                    out[1] = invSamples * ps.acc1;
                    out[2] = invSamples * ps.acc2;
                }
                const uint32_t n = __popcll(needMask);
                uint32_t g = 0;
                if (lane == 0)
                    g = __hip_atomic_fetch_add(args.pixelCounter, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g = __builtin_amdgcn_readfirstlane(g);
                if (needPixel) {
                    const uint32_t idx = g + __popcll(needMask & ((1ull << lane) - 1ull));
                    if (idx < args.blockSize && g <= 0xffffffffu - 64u) {
                        if (args.tiled) {
                            const uint32_t tilesPerRow = args.width >> 3;
                            const uint32_t tile = idx >> 6, in = idx & 63u;
                            const uint32_t tx = tile % tilesPerRow, ty = tile / tilesPerRow;
                            pixel = args.blockStart + ((ty << 3) + (in >> 3)) * args.width + (tx << 3) + (in & 7u);
                        } else {
                            pixel = args.blockStart + idx;
                        }
                        pathStateInit(ps, pixel, args.width);
                        next = blockNew<F>(fa, ps);
                        dead = next != NEXT_TRACE; /* cannot happen (at least one sample); never lose a slot */
                    } else {
                        dead = true;
                    }
                }
                const uint32_t nDead = __popcll(__ballot(dead));
                if (nDead > 0 && lane == 0)
                    __hip_atomic_fetch_add(ctl + WF_DONE_SLOTS, nDead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (stats) {
                    sched[9]++;
                    sched[10] += n;
                }
            }
            const bool posting = mine && !dead && next == NEXT_TRACE;
            if (posting) {
                if constexpr (GSTATE) {
                    float4* rp = grec + (size_t)s * WF_GRECORD4;
                    const uint32_t newMeta = (ps.pathComponent & 0xffffu) | ((uint32_t)ps.rayKind << 16);
                    rp[0] = make_float4(ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, ps.ray.d.x);
                    rp[1] = make_float4(ps.ray.d.y, ps.ray.d.z, __uint_as_float(newMeta), __uint_as_float(pixel));
                    /* q2 and q3.xy are the traversal's to write */
                    reinterpret_cast<float2*>(rp + 3)[1] = make_float2(__uint_as_float(ps.sampleIndex), __uint_as_float(ps.chosenPrim));
                    rp[4] = make_float4(__uint_as_float(ps.prng.s0), __uint_as_float(ps.prng.s1), __uint_as_float(ps.prng.s2), __uint_as_float(ps.prng.s3));
                    rp[5] = make_float4(ps.acc0, ps.acc1, ps.acc2, 0.0f);
                    rp[6] = make_float4(ps.ray.ri.x, ps.ray.ri.y, ps.ray.ri.z, ps.ray.ri.w);
                    rp[7] = make_float4(ps.att.x, ps.att.y, ps.att.z, ps.att.w);
                    rp[8] = make_float4(ps.opl.x, ps.opl.y, ps.opl.z, 0.0f);
                    rp[9] = make_float4(ps.nextAtt.x, ps.nextAtt.y, ps.nextAtt.z, ps.nextAtt.w);
                    rp[10] = make_float4(ps.neeFactor.x, ps.neeFactor.y, ps.neeFactor.z, ps.neeFactor.w);
                    rp[11] = make_float4(ps.srDir.x, ps.srDir.y, ps.srDir.z, 0.0f);
                } else {
                    fld(WF_F_PRNG + 0, s) = ps.prng.s0;
                    fld(WF_F_PRNG + 1, s) = ps.prng.s1;
                    fld(WF_F_PRNG + 2, s) = ps.prng.s2;
                    fld(WF_F_PRNG + 3, s) = ps.prng.s3;
                    fldF(WF_F_ACC + 0, s) = ps.acc0;
                    fldF(WF_F_ACC + 1, s) = ps.acc1;
                    fldF(WF_F_ACC + 2, s) = ps.acc2;
                    fld(WF_F_PIXEL, s) = pixel;
                    fld(WF_F_SAMPLE, s) = ps.sampleIndex;
                    fld(WF_F_META, s) = (ps.pathComponent & 0xffffu) | ((uint32_t)ps.rayKind << 16);
                    fldF(WF_F_ORG + 0, s) = ps.ray.o.x;
                    fldF(WF_F_ORG + 1, s) = ps.ray.o.y;
                    fldF(WF_F_ORG + 2, s) = ps.ray.o.z;
                    fldF(WF_F_DIR + 0, s) = ps.ray.d.x;
                    fldF(WF_F_DIR + 1, s) = ps.ray.d.y;
                    fldF(WF_F_DIR + 2, s) = ps.ray.d.z;
                    fldF(WF_F_RI + 0, s) = ps.ray.ri.x;
                    fldF(WF_F_RI + 1, s) = ps.ray.ri.y;
                    fldF(WF_F_RI + 2, s) = ps.ray.ri.z;
                    fldF(WF_F_RI + 3, s) = ps.ray.ri.w;
                    fldF(WF_F_ATT + 0, s) = ps.att.x;
                    fldF(WF_F_ATT + 1, s) = ps.att.y;
                    fldF(WF_F_ATT + 2, s) = ps.att.z;
                    fldF(WF_F_ATT + 3, s) = ps.att.w;
                    fldF(WF_F_OPL + 0, s) = ps.opl.x;
                    fldF(WF_F_OPL + 1, s) = ps.opl.y;
                    fldF(WF_F_OPL + 2, s) = ps.opl.z;
                    fldF(WF_F_NEXTATT + 0, s) = ps.nextAtt.x;
                    fldF(WF_F_NEXTATT + 1, s) = ps.nextAtt.y;
                    fldF(WF_F_NEXTATT + 2, s) = ps.nextAtt.z;
                    fldF(WF_F_NEXTATT + 3, s) = ps.nextAtt.w;
                    fldF(WF_F_NEEFACTOR + 0, s) = ps.neeFactor.x;
                    fldF(WF_F_NEEFACTOR + 1, s) = ps.neeFactor.y;
                    fldF(WF_F_NEEFACTOR + 2, s) = ps.neeFactor.z;
                    fldF(WF_F_NEEFACTOR + 3, s) = ps.neeFactor.w;
                    fldF(WF_F_SRDIR + 0, s) = ps.srDir.x;
                    fldF(WF_F_SRDIR + 1, s) = ps.srDir.y;
                    fldF(WF_F_SRDIR + 2, s) = ps.srDir.z;
                    fld(WF_F_CHOSEN, s) = ps.chosenPrim;
                }
            }
            wfRelease();
            wfPush(ringT, ctl + WF_T_TAIL, abortFlag, posting, s, lane, RING - 1);
            continue;
        }

        /* ---- traversal: refill idle contexts, a quarter of the wave at a time ---- */
        if (travWave && tr > 0u && (nAct == 0 || 64 - nAct >= 16)) {
            const bool idle = tst == T_IDLE;
            const uint32_t s = wfPop(ringT, ctl + WF_T_HEAD, ctl + WF_T_TAIL, abortFlag, idle, (uint32_t)__popcll(__ballot(idle)), lane, RING - 1);
            const uint32_t nGot = __popcll(__ballot(s != WF_EMPTY));
            if (nGot > 0 && lane == 0)
                __hip_atomic_fetch_add(ctl + WF_ACTIVE, nGot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (s != WF_EMPTY) {
                wfAcquire();
                tslot = s;
                if constexpr (GSTATE) {
                    const float4* rp = grec + (size_t)s * WF_GRECORD4;
                    const float4 q0 = rp[0];
                    const float2 q1 = *reinterpret_cast<const float2*>(rp + 1);
                    org = mk3(q0.x, q0.y, q0.z);
                    aux = rayAux(mk3(q0.w, q1.x, q1.y));
                } else {
                    org = mk3(fldF(WF_F_ORG + 0, s), fldF(WF_F_ORG + 1, s), fldF(WF_F_ORG + 2, s));
                    aux = rayAux(mk3(fldF(WF_F_DIR + 0, s), fldF(WF_F_DIR + 1, s), fldF(WF_F_DIR + 2, s)));
                }
                node = 0;
                amax = k_maxval;
                best.prim = NO_HIT;
                tst = T_NODE;
            }
        }
        const int nWalk = __popcll(__ballot(tst == T_NODE || tst == T_LEAF));
        if (nWalk == 0) {
            /* nothing to traverse and no batch worth taking */
            if (wfLoad(ctl + WF_DONE_SLOTS) >= slots)
                break;
            __builtin_amdgcn_s_sleep(8);
            patience++;
            if (++idleSpins > WF_SPIN_LIMIT) {
                wfStore(abortFlag, 1u);
                break;
            }
            continue;
        }
        idleSpins = 0;
        patience = 0;
        if (stats)
            sched[0]++;
        /* walk: NODE steps and LEAF tests until another quarter of the wave has finished (then
         * results are handed back and the contexts refilled), at the latest after 64 steps */
        const int leaveBelow = nWalk > 16 ? nWalk - 16 : 0;
        for (int steps = 0; steps < 64; steps++) {
            const int nNode = __popcll(__ballot(tst == T_NODE));
            const int nLeaf = __popcll(__ballot(tst == T_LEAF));
            if (nNode + nLeaf <= leaveBelow)
                break;
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
                    tst = node >= nodeCount ? (int)T_FIN : (int)T_NODE;
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
                    tst = toLeaf ? (int)T_LEAF : ((!toLeaf && node >= nodeCount) ? (int)T_FIN : (int)T_NODE);
                }
            }
        }
    }
    if (stats && lane == 0) {
        for (int i = 0; i < 11; i++)
            atomicAdd(args.schedStats + i, sched[i]);
    }
    /* a launch that had to abort must not look like a finished frame */
    if (wfLoad(abortFlag) && args.status && lane == 0)
        atomicExch(args.status, 1u);
}

void launchWfBasicLds(const KernelArgs& args, dim3 grid, size_t ldsBytes, hipStream_t stream);
void launchWfBasic(const KernelArgs& args, dim3 grid, size_t ldsBytes, hipStream_t stream);
void launchWfFull(const KernelArgs& args, dim3 grid, size_t ldsBytes, hipStream_t stream);
void launchWfgBasicLds(const KernelArgs& args, dim3 grid, size_t ldsBytes, hipStream_t stream);
void launchWfgFull(const KernelArgs& args, dim3 grid, size_t ldsBytes, hipStream_t stream);

inline size_t wfgLdsBytes(uint32_t sceneBytes) { return size_t(sceneBytes) + size_t(2 * WF_GRING + 16) * 4; }

} /* namespace wptk */

#endif
