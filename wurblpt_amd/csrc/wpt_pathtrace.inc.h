/*
 * wpt_pathtrace.inc.h -- the gfx950 path-tracing kernel (template; one translation unit per
 * instantiation, wpt_k_*.hip, so that the variants compile in parallel).
 *
 * Shape of the kernel, MI355X first:
 *
 *  - One lane owns one pixel and walks that pixel's whole sample sequence, because the
 *    reference seeds ONE Prng per pixel and consumes it serially over all samples
 *    (wurblpt.hpp:342-366).  Pixels are the only parallel axis; a wave is an 8x8 pixel tile.
 *
 *  - Every lane is a small state machine (NEW ray -> NODE steps <-> LEAF tests -> SHADE or
 *    NEE-END -> ...).  The 64 lanes of a wave are in different states at any time, so the wave
 *    runs its own scheduler: each round it counts the lanes per state with ballots and runs
 *    either the traversal loop (left as soon as too few lanes remain in it) or one long round
 *    that serves the SHADE, NEE-END and NEW lanes together.  Lanes in other states simply wait.
 *    Each lane still executes its own operations in the reference's order, so results do not
 *    change; only SIMD utilisation does (v0 without the scheduler: 18 % active lanes).
 *
 *  - Registers hold the ray and the walk (origin, direction, reciprocals, shear, node, bound, best
 *    candidate: 24 words).  The rest of the path's state (32 words: generator, accumulators,
 *    attenuations, the continuation behind a next-event ray) lives in LDS, eight 16-byte slots per
 *    lane (wpt_blocks.h): 32 KiB per 256-thread workgroup, so four workgroups per CU still fit, and
 *    the long round no longer spills the waiting lanes' words to scratch memory.
 *
 *  - BVH traversal is stackless.  The reference walks its depth-first node array with a stack,
 *    left child first (bvh.hpp:277-311); the node a pop returns to is always the first node behind
 *    the current subtree in depth-first order, so each device node carries that node's index
 *    ("skip") and an inner node the index of its first child: box hit & inner -> first child;
 *    leaf -> test it, then skip; otherwise -> skip.  Same visiting order, no stack, and the nodes
 *    may be STORED in any order (wpt_capi.hip lays the top of a large tree out level by level).
 *
 *  - Small scenes (nodes + triangle positions up to 20 KiB, e.g. the Cornell box: 4 KiB) are
 *    copied into LDS once per workgroup and traversed from there; larger scenes are fetched
 *    from HBM/L2 as two dwordx4 per node and three per triangle.
 *
 *  - Shading data (96 B per triangle) is read once per ray, after traversal.
 *  - No MFMA: there is no dense contraction anywhere on this path.
 */
#ifndef WPT_PATHTRACE_INC_H
#define WPT_PATHTRACE_INC_H

#include <hip/hip_runtime.h>

#include "../../include/wurblpt_hip.h"
#include "wpt_blocks.h"

namespace wptk {

using namespace wptd;

constexpr int WG = 256; /* threads per workgroup: 4 waves, one per SIMD */
/* device node, word 7: a triangle index (< 2^31), PRIM_SPHERE | sphere index (< 2^30), or NODE_CHILD | index of the
 * first child (an empty node: NODE_CHILD | its own skip link, so that entering it is the same as skipping it) */
constexpr uint32_t NODE_CHILD = 0xc0000000u;
constexpr uint32_t NODE_INDEX_MASK = 0x3fffffffu;
/* dynamic LDS of a workgroup: the tables of expf / powf (wpt_math.h, 512 B), the paths' cold words (32 KiB), then the
 * scene if it is small */
constexpr uint32_t TABLE_BYTES = WPT_MATH_TABLE_WORDS * 8;
constexpr uint32_t COLD_BYTES = TABLE_BYTES + SLOT_COUNT * WG * 16;
/* with the scene behind them three workgroups still fit into a CU's 160 KiB */
constexpr uint32_t LDS_SCENE_MAX_BYTES = 20 * 1024;
/* a quarter of a compute unit's 160 KiB: what a workgroup may use where four of them are to share the unit (the material
 * records join the scene in LDS where they still fit into it) */
constexpr uint32_t LDS_BYTES_PER_WORKGROUP_AT_FOUR = 160 * 1024 / 4;
/* the material records are copied to LDS quadword by quadword and read there through a generic pointer */
static_assert(sizeof(wpt_material) % 16 == 0 && alignof(wpt_material) <= 16, "wpt_material must be a whole number of quadwords");
struct KernelArgs {
    SceneView sv;
    wpt_camera cam;
    wpt_params par;
    uint32_t width, height, samplesSqrt;
    float invWidth, invHeight, invSamplesSqrt, invSamples; /* 1.0f / (float)..., divided on the host */
    uint32_t blockStart, blockSize;
    uint32_t tiled; /* 1: a wave covers an 8x8 pixel tile (block is whole rows, multiple of 8) */
    /* bands: with bandStride > 0 the launch covers the bands of bandPixels consecutive pixels
     * whose index is bandFirst, bandFirst + bandStride, ... -- one rank's interleaved share of the frame in one launch;
     * blockStart is 0 and blockSize the number of lanes (pixels behind the end of the frame stay idle) */
    uint32_t bandPixels, bandFirst, bandStride;
    uint32_t leaveEighths; /* scheduler: leave the NODE loop when fewer than this many eighths of the entering lanes remain */
    uint32_t heavyMin;     /* scheduler: lanes a long block needs before it runs */
    uint32_t leafBias;     /* scheduler: leaf tests run when waiting lanes * leafBias >= walking lanes * 8 */
    uint32_t waitBelow;    /* scheduler: a kind of material with fewer lanes than this in a long round stands back once (0 = never) */
    uint32_t fuse;         /* scheduler: 1 = one long round serves SHADE, NEE-END and NEW lanes together */
    uint32_t shadowWalksEnd; /* 1: the walk of a light ray towards the environment ends at its first accepted hit (not in counting launches) */
    float* frame;
    /* Pixel pool (or NULL): lanes whose pixel is finished take the next lane index of the launch from this counter, which
     * starts at the number of lanes launched.  A launch then is as many workgroups as the GPU holds at once, and a wave
     * keeps its 64 lanes at work until the launch runs out of pixels instead of until its slowest pixel ends. */
    uint32_t* pool;
    /* Two passes over a frame (or NULL / 0): the first renders rowStop rows of every pixel's strata, stores what the pixel
     * carries on (generator, sum, next stratum) and the time it took; the second takes the pixels in `order` (the tiles
     * that took longest first, wpt_k_order.hip: the pixels that end the launch are the short ones) and goes on where the
     * first stopped.  A pixel's samples stay one sequence from one generator, so the frame is the same bit for bit. */
    uint32_t rowStop;             /* rows of strata to render in this launch; samplesSqrt: to the end */
    float4* carry;                /* per pixel of the frame: prng slot, acc slot */
    uint32_t* cost;               /* per pixel of the frame: shader clock of the first pass */
    const uint32_t* order;        /* second pass: pixels in the order they are handed out */
    const uint32_t* orderCount;   /* second pass: entries of `order` */
    uint32_t cuCount; /* for the launchers: compute units of the device */
    uint32_t materialsInLds; /* scene in LDS: the material records are there too */
    wpt_counters* counters;
    unsigned long long* schedStats; /* COUNT builds: 16 scheduler statistics, or NULL */
};

/* lane index of the launch -> pixel; false: no pixel behind this index */
WPT_D bool lanePixel(const KernelArgs& args, uint32_t gid, uint32_t& pixel)
{
    bool inBlock = gid < args.blockSize;
    if (args.tiled) {
        const uint32_t tilesPerRow = args.width >> 3;
        const uint32_t tile = gid >> 6, lane = gid & 63u;
        const uint32_t tx = tile % tilesPerRow;
        uint32_t ty = tile / tilesPerRow;
        if (args.bandStride) {
            /* tile rows of this launch -> tile rows of the frame (bands are whole groups of 8 rows here) */
            const uint32_t tileRowsPerBand = args.bandPixels / (args.width << 3);
            ty = (args.bandFirst + (ty / tileRowsPerBand) * args.bandStride) * tileRowsPerBand + ty % tileRowsPerBand;
        }
        pixel = args.blockStart + ((ty << 3) + (lane >> 3)) * args.width + (tx << 3) + (lane & 7u);
    } else if (args.bandStride) {
        pixel = (args.bandFirst + (gid / args.bandPixels) * args.bandStride) * args.bandPixels + gid % args.bandPixels;
    } else {
        pixel = args.blockStart + gid;
    }
    if (args.bandStride && pixel >= args.width * args.height)
        inBlock = false; /* the last band may be shorter */
    if (!inBlock)
        pixel = args.blockStart;
    return inBlock;
}

/* lane states, in scheduling priority order for ties */
enum { S_NODE = 0, S_LEAF = 1, S_SHADE = 2, S_NEEEND = 3, S_NEW = 4, S_DONE = 5, S_START = 6 /* within a long round: has a ray to start */ };

/* ---- the wide walk (WIDE kernels; scenes fetched from HBM) ----
 * The binary tree collapsed by one level at upload (wpt_capi.hip, SceneView::wideNodes): a wide node holds the boxes of a
 * node's up to four grandchildren (a child that is a leaf stands for itself) in the order BVH::hit comes to them
 * (bvh.hpp:277-311), and one step tests all four from one 128-byte line.  A child whose box the ray passes through under the
 * bound of that moment waits for its turn with its entry distance -- the next one in registers, the others on a small stack --
 * and is admitted at its turn if that distance is still within the bound, which is AABB::mayHit (aabb.hpp:70-86) under the
 * bound of its turn as long as (a) no slab distance is NaN and (b) the bound has not grown in between.  The inner nodes
 * that disappear decide nothing of their own: a child's box lies within its parent's (checked at upload) and the slab
 * arithmetic is monotone, so a child that passes implies the parent the reference tested before it.  Where (a) or (b) fails
 * the lane walks the binary tree instead, which IS the reference's walk:
 *   (a) a slab distance is NaN only for 0 * inf or inf - inf: a ray with a zero, infinite or NaN direction component, or an
 *       origin that is not finite (the boxes are finite: checked at upload), takes the binary walk from its start;
 *   (b) a hit is accepted by one comparison and its distance stored by another (hitable_triangle.hpp:283-296), so the bound
 *       can GROW by an ulp at a hit (one ray in 200 000 on the Sponza-class scene, tests/test_wide_walk.py); a lane that
 *       sees that starts its ray again from the root in the binary walk.
 * Checked on the CPU leaf test for leaf test against BVH::hit (oracle/wpt_oracle.cpp::bvhTraverseWide) and on the GPU bit for
 * bit against the oracle.  The stack cannot overflow: wpt_scene_upload offers the wide form only for trees whose worst case
 * fits WIDE_STACK entries. */
constexpr uint32_t WIDE_STACK = 96;          /* pending children per lane (8 bytes each, scratch memory): the 10 M triangle scene's worst case is 66 */
constexpr uint32_t WIDE_NONE = 0xffffffffu;  /* no child (also: an empty entry of a wide node) */
constexpr int RAY_WALK_BINARY = 0x80;        /* in RayAux::k: this ray walks the binary tree although its slab distances are numbers (its bound has grown) */

template<uint32_t F, bool COUNT, bool LDSSCENE, int OCC, bool WIDE = false>
__global__ __launch_bounds__(WG, OCC) void wpt_pathtrace(const KernelArgs args)
{
    static_assert(!WIDE || (!LDSSCENE && !COUNT), "the wide walk: product kernels that fetch the scene from HBM (counting launches walk like the reference)");
    /* node prefetch: for scenes in HBM (Sponza-class frame 3 % faster); not from LDS, where the fetch is short and the
     * registers that hold the node ahead lengthen every step (Cornell 4 % slower) */
    constexpr bool PREFETCH = !LDSSCENE && !WIDE;
    /* Node steps per look at the lane counts.  The look itself (two ballots, their counts, the leave and leaf decisions:
     * some twenty scalar instructions and two branches in every lane's way) costs a wave as much issue time as half a
     * node step.  From LDS a step is short, and taking up to three in a row before looking again gave 852 against 799
     * Msamples/s on the Cornell frame (2: 839, 4: 841, 6: 836, 8: 791: lanes that reach a leaf wait out the rest); from
     * HBM the steps are memory round trips and nothing is gained. */
    constexpr int STEPS = LDSSCENE ? 3 : 1;
    /* Lanes at the end of a light ray that the walk serves itself (0: they wait for the long round).  Kernels that fetch the scene
     * from HBM wait for memory, not for instruction issue: there a lane that walks on sooner is worth the extra run of the short
     * block (Sponza-class 152.7 -> 158.0 Msamples/s with 3 or 4 lanes, 157.3 / 156.8 with 8 / 12, 155.1 with 2; measured BRDFs in
     * the single kernel 106.1 -> 112.5; 10 M triangles 67.97 -> 68.09).  With the scene in LDS the frame is bound by the
     * instructions its waves issue and the same costs (Cornell 1055.6 against 1038.7 / 1049.1 / 1036.0 with 8 / 16 / 24). */
    constexpr int NEE_IN_WALK = LDSSCENE ? 0 : 4;
    /* [ math tables ][ cold path words: SLOT_COUNT x WG float4 ][ LDSSCENE: nodes, triangle positions ] */
    extern __shared__ float4 lds[];
    float4* const ldsCold = lds + TABLE_BYTES / 16;
    float4* const ldsScene = ldsCold + SLOT_COUNT * WG;
#if defined(WPT_MATH_TABLES_IN_LDS) && defined(__HIP_DEVICE_COMPILE__)
    wptm::tables_to_lds(threadIdx.x);
#endif

    /* the kernel with the scene in LDS points its own copy of the scene view at the material records there; the others
     * read the launch arguments where they lie (a copy costs the all-features kernel 2 %) */
    SceneView svInLds = args.sv;
    const SceneView& sv = LDSSCENE ? static_cast<const SceneView&>(svInLds) : args.sv;
    const wpt_params& par = args.par;
    const uint32_t nodeCount = sv.nodeCount;

    if (LDSSCENE) {
        /* nodes (2 x float4 each) followed by the triangle positions (3 x float4 each) */
        const uint32_t n4 = 2 * nodeCount, t4 = 3 * sv.triCount;
        for (uint32_t i = threadIdx.x; i < n4; i += WG)
            ldsScene[i] = sv.nodes[i];
        for (uint32_t i = threadIdx.x; i < t4; i += WG)
            ldsScene[n4 + i] = sv.triGeom[i];
        if (args.materialsInLds) {
            /* the material record is what a hit's shading waits for first: fetched from LDS through a generic pointer */
            const uint32_t m4 = sv.materialCount * (uint32_t)(sizeof(wpt_material) / 16);
            const float4* from = reinterpret_cast<const float4*>(sv.materials);
            for (uint32_t i = threadIdx.x; i < m4; i += WG)
                ldsScene[n4 + t4 + i] = from[i];
            svInLds.materials = reinterpret_cast<const wpt_material*>(ldsScene + n4 + t4);
        }
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

    auto pixelOf = [&](uint32_t gid, uint32_t& pixel) -> bool { return lanePixel(args, gid, pixel); };
    const bool firstPass = args.rowStop < args.samplesSqrt;
    const uint32_t laneLimit = args.order ? *args.orderCount : args.blockSize; /* indices a lane may take */
    FrameArgs fa;
    fa.cam = args.cam;
    fa.par = args.par;
    fa.width = args.width;
    fa.height = args.height;
    fa.samplesSqrt = args.samplesSqrt;
    fa.invWidth = args.invWidth;
    fa.invHeight = args.invHeight;
    fa.invSamplesSqrt = args.invSamplesSqrt;

    /* ---- per-lane state: the pixel's path (wpt_blocks.h; its cold words in LDS) and the traversal registers ---- */
    PathLds<WG> ps;
    ps.base = ldsCold + threadIdx.x;
    /* the lane takes the pixel behind index `at` of the launch; false: there is none */
    auto startPixel = [&](uint32_t at) -> bool {
        uint32_t pixel = args.blockStart;
        bool have;
        if (args.order) {
            have = at < laneLimit;
            if (have)
                pixel = args.order[at];
        } else {
            have = pixelOf(at, pixel);
        }
        pathStateInit(ps, pixel, pixel % args.width, pixel / args.width);
        if (have && args.order) {
            /* where the first pass stopped */
            ps.base[SLOT_PRNG * WG] = args.carry[2 * (size_t)pixel];
            ps.base[SLOT_ACC * WG] = args.carry[2 * (size_t)pixel + 1];
        }
        if (have && firstPass)
            args.cost[pixel] = (uint32_t)clock64();
        return have;
    };
    const bool inBlock = startPixel(blockIdx.x * WG + threadIdx.x);
    LaneCounters lc = LANE_COUNTERS_ZERO;
    /* wave-level scheduler statistics (COUNT builds): rounds and lane counts per state */
    unsigned long long sched[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    int state = inBlock ? S_NEW : S_DONE;
    bool poolDry = false; /* wave-uniform */
    RayAux aux = rayAux(ps.d);
    uint32_t node = 0, leafPrim = 0;
    float amax = k_maxval;
    /* WIDE: the child whose turn is next (reference, entry distance) waits in registers, so that a step starts with its node's
     * fetch and not with a load from the stack; the others wait on the stack in the reference's order */
    uint2 pend[WIDE ? WIDE_STACK : 1];
    uint32_t sp = 0, curRef = WIDE_NONE;
    float curEntry = 0.0f;
    Candidate best;
    best.prim = NO_HIT;
    best.a = best.invDet = best.U = best.V = best.W = 0.0f;

    /* start the traversal of the ray ps.o, ps.d */
    auto beginRay = [&]() {
        aux = rayAux(ps.d);
        node = 0;
        amax = k_maxval;
        best.prim = NO_HIT;
        state = S_NODE;
        /* rays for which a slab distance can be NaN (a zero direction component, say) are rare; only they test for it per box,
         * and in WIDE kernels they walk the binary tree */
        if (rayMayNan(ps.o, aux.inv))
            aux.k |= RAY_MAY_NAN;
        if (WIDE) {
            /* the root's wide node, admitted under any bound (the root's own box decides nothing its children do not) */
            curRef = NODE_CHILD | 0u;
            curEntry = 0.0f;
            sp = 0;
        }
        if (COUNT)
            lc.rays++;
    };
    /* state after the walk has left the tree */
    auto endOfRayState = [&]() { return ps.rayKind == RAY_PATH ? (int)S_SHADE : (int)S_NEEEND; };
    /* what a block of wpt_blocks.h asks for next.  Rays start together at the end of the long round: the reciprocals and
     * the shear of a direction (three divisions and the axis choice, some sixty instructions) then run once for the
     * SHADE, NEE-END and NEW lanes of the round, not once behind each of their blocks. */
    auto afterBlock = [&](int next) {
        if (next == NEXT_TRACE)
            state = S_START;
        else if (next != NEXT_WAIT) /* a waiting lane keeps its state and its hit */
            state = next == NEXT_NEW ? (int)S_NEW : (int)S_DONE;
    };

    /* idle lanes take the next pixels of the launch: one atomic per wave */
    auto fromPool = [&]() {
        if (!COUNT && args.pool && !poolDry) {
            const unsigned long long idle = __ballot(state == S_DONE);
            if (idle != 0) {
                const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                const uint32_t want = (uint32_t)__popcll(idle);
                const int leader = __ffsll((long long)idle) - 1;
                uint32_t first = 0;
                if ((int)lane == leader)
                    first = atomicAdd(args.pool, want);
                first = (uint32_t)__builtin_amdgcn_readlane((int)first, leader);
                poolDry = first + want >= laneLimit; /* the counter only grows: nothing behind it for this wave */
                if (state == S_DONE) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                    if (startPixel(first + rank))
                        state = S_NEW;
                }
            }
        }
    };

    for (;;) {
        /* ---- the wave's scheduler ----
         * Traversal (NODE steps and LEAF tests) is one block with its own inner policy; the long
         * blocks (SHADE, NEE-END, NEW) run when they are well filled, or when no traversal work
         * is left in the wave.  Waiting lanes lose nothing but time: every lane still executes
         * its own operations in order. */
        sec<COUNT>(lc, SEC_LOOK);
        const int cTrav = __popcll(__ballot(state == S_NODE || state == S_LEAF));
        const int cShade = __popcll(__ballot(state == S_SHADE));
        const int cNee = __popcll(__ballot(state == S_NEEEND));
        const int cNew = __popcll(__ballot(state == S_NEW));
        if ((cTrav | cShade | cNee | cNew) == 0)
            break;
        int pick;
        const bool fused = args.fuse != 0;
        {
            /* at least 1: a block must never be picked with no lane in it */
            const int heavyMin = (int)args.heavyMin < 1 ? 1 : (int)args.heavyMin;
            if (fused) {
                /* one long round for every lane that is not traversing: fewer lanes wait for
                 * "their" block to fill up, at the price of running up to three code sections */
                pick = (cShade + cNee + cNew >= heavyMin || cTrav == 0) ? (int)S_SHADE : (int)S_NODE;
            } else if (cShade >= heavyMin) {
                pick = S_SHADE;
            } else if (cNee >= heavyMin) {
                pick = S_NEEEND;
            } else if (cNew >= heavyMin) {
                pick = S_NEW;
            } else if (cTrav != 0) {
                pick = S_NODE;
            } else {
                int most = cShade;
                pick = S_SHADE;
                if (cNee > most) { pick = S_NEEEND; most = cNee; }
                if (cNew > most) { pick = S_NEW; most = cNew; }
            }
        }
        long long tBlock = 0;
        if (COUNT)
            tBlock = clock64();
        if (pick == S_NODE) {
            /* Leave when fewer than leaveEighths/8 of the entering lanes are still traversing
             * (never below 1: the loop must end when no lane is left in it). */
            int leaveBelow = (cTrav * (int)args.leaveEighths + 7) >> 3;
            leaveBelow = leaveBelow < 1 ? 1 : leaveBelow;
            if (COUNT)
                sched[0]++;
            /* the node a lane will test next is known one iteration ahead: its two quadwords are requested at the end
             * of the iteration before, so that the fetch runs behind the loop's ballots and branches */
            float4 pn0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pn1 = pn0;
            /* does any ray of the wave need its slab distances tested for NaN? (rays start in the long round only) */
            bool nanPossible = sv.boxesMayBeNan != 0 || __ballot((aux.k & RAY_MAY_NAN) != 0 && (state == S_NODE || state == S_LEAF)) != 0;
            /* AABB::mayHit + the stackless form of BVH::hit's walk: one node step of the lanes in state NODE (WIDE: of those among
             * them that walk the binary tree) */
                auto binaryStep = [&]() {
                if (COUNT)
                    lc.nodes++;
                const float4 n0 = PREFETCH ? pn0 : node4(2 * node), n1 = PREFETCH ? pn1 : node4(2 * node + 1);
                const uint32_t skip = __float_as_uint(n1.z);
                const uint32_t word = __float_as_uint(n1.w);
                bool hit = boxTest<false>(nodeLo(n0, n1), nodeHi(n0, n1), ps.o, aux.inv, par.min_hit_distance, amax);
                if (nanPossible) /* wave-uniform, rarely true */
                    hit = boxTest<true>(nodeLo(n0, n1), nodeHi(n0, n1), ps.o, aux.inv, par.min_hit_distance, amax);
                /* select form of: hit & inner -> first child; hit & leaf -> test it, then skip; else -> skip */
                const bool inner = word >= NODE_CHILD;
                const bool toLeaf = hit && !inner;
                leafPrim = toLeaf ? word : leafPrim;
                node = (hit && inner) ? (word & NODE_INDEX_MASK) : skip;
                state = toLeaf ? (int)S_LEAF : (int)S_NODE;
                if (!toLeaf && node >= nodeCount)
                    state = endOfRayState();
                if (PREFETCH && state == S_NODE) {
                    pn0 = node4(2 * node);
                    pn1 = node4(2 * node + 1);
                }
            };
            if (PREFETCH && state == S_NODE) {
                pn0 = node4(2 * node);
                pn1 = node4(2 * node + 1);
            }
            for (;;) {
                sec<COUNT>(lc, SEC_LOOK_INNER);
                const int nNode = __popcll(__ballot(state == S_NODE));
                const int nLeaf = __popcll(__ballot(state == S_LEAF));
                if (nNode + nLeaf < leaveBelow)
                    break;
                /* The end of a light ray is a short block (its contribution, roulette, the continuation's start): lanes that
                 * wait for it do not have to wait for the long round -- once NEE_IN_WALK of them are there, the walk serves them
                 * and they walk on with the rest.  (The same block as in the long round; a lane's order of operations is its own.) */
                if (NEE_IN_WALK > 0 && !COUNT) {
                    const int nNee = __popcll(__ballot(state == S_NEEEND));
                    if (nNee >= NEE_IN_WALK) {
                        if (state == S_NEEEND)
                            afterBlock(blockNeeEnd<F>(sv, par, tri4, ps, best));
                        if (state == S_START) {
                            beginRay();
                            if (PREFETCH) {
                                pn0 = node4(2 * node);
                                pn1 = node4(2 * node + 1);
                            }
                        }
                        nanPossible = nanPossible || __ballot((aux.k & RAY_MAY_NAN) != 0 && state == S_NODE) != 0; /* rays have started */
                        continue;
                    }
                }
                /* a leaf test is ~3 node steps long; it runs once enough lanes wait for it
                 * (leafBias/8 of the walking lanes), because waiting lanes thin out the walk */
                if (nLeaf * (int)args.leafBias >= nNode * 8 && nLeaf > 0) {
                    if (COUNT) {
                        sched[3]++;
                        sched[4] += nLeaf;
                    }
                    sec<COUNT>(lc, SEC_LEAF_TEST, state == S_LEAF);
                    if (state == S_LEAF) {
                        /* HitableTriangle::hit, candidate part (hitable_triangle.hpp:189-271); the walk already knows where
                         * it goes on (a leaf's subtree is the leaf itself) */
                        if (COUNT)
                            lc.leaves++;
                        Candidate c;
                        bool accepted;
                        if ((F & FEAT_SPHERES) && (leafPrim & PRIM_SPHERE)) {
                            /* HitableSphere::hit (hitable_sphere.hpp:104-147) */
                            c.invDet = c.U = c.V = c.W = 0.0f;
                            accepted = sphereTest(sphereNow<F>(sv, ps, sv.spheres[leafPrim & ~PRIM_SPHERE]), ps.o, ps.d, par.min_hit_distance, amax, c.a);
                        } else {
                            const float4 g0 = tri4(3 * leafPrim), g1 = tri4(3 * leafPrim + 1), g2 = tri4(3 * leafPrim + 2);
                            f3 v0 = mk3(g0.x, g0.y, g0.z), v1 = mk3(g1.x, g1.y, g1.z), v2 = mk3(g2.x, g2.y, g2.z);
                            if ((F & FEAT_ANIM) && (__float_as_uint(g2.w) & WPT_TRI_ANIMATE)) {
                                /* the instance moves: its corners at the ray's time (hitable_triangle.hpp:209-218) */
                                const float* animationM = animationMatrix(sv, ps, sv.instances[__float_as_uint(g0.w)].animation);
                                v0 = animatePoint(animationM, v0);
                                v1 = animatePoint(animationM, v1);
                                v2 = animatePoint(animationM, v2);
                            }
                            accepted = triangleTest(v0, v1, v2, ps.o, aux, par.min_hit_distance, amax, c);
                        }
                        /* WIDE: a hit whose stored distance lies beyond the bound it was accepted under (the two are separate
                         * comparisons in the reference) makes the bound grow: children dropped under the smaller bound may be
                         * due after all, so this ray starts again from the root in the binary walk */
                        const bool grown = WIDE && accepted && !(aux.k & (RAY_MAY_NAN | RAY_WALK_BINARY)) && !(c.a <= amax);
                        if (accepted) {
                            c.prim = leafPrim;
                            best = c;
                            amax = c.a;
                        }
                        if (WIDE && !(aux.k & (RAY_MAY_NAN | RAY_WALK_BINARY)))
                            state = curRef == WIDE_NONE ? endOfRayState() : (int)S_NODE;
                        else
                            state = node >= nodeCount ? endOfRayState() : (int)S_NODE;
                        /* A light ray towards the environment asks one thing: is anything in the way (blockNeeEnd,
                         * wurblpt.hpp:240-250).  Up to a walk's first accepted hit the bound is the ray's own, so every box
                         * and leaf decision is the reference's, and the answer is known there: the walk ends.  (Counting
                         * launches are given shadowWalksEnd = 0 and walk on, as the reference does: their numbers are its numbers.) */
                        if ((F & FEAT_ENVMAP) && args.shadowWalksEnd && accepted && ps.rayKind == RAY_NEE_ENV) {
                            state = S_NEEEND;
                        } else if (grown) {
                            aux.k |= RAY_WALK_BINARY;
                            node = 0;
                            amax = k_maxval;
                            best.prim = NO_HIT;
                            state = S_NODE;
                        }
                        if (PREFETCH && state == S_NODE) {
                            pn0 = node4(2 * node);
                            pn1 = node4(2 * node + 1);
                        }
                    }
                } else {
                    if (COUNT) {
                        sched[1]++;
                        sched[2] += nNode;
                    }
                    if constexpr (WIDE) {
                        if (state == S_NODE && !(aux.k & (RAY_MAY_NAN | RAY_WALK_BINARY))) {
                            /* The child whose turn it is: admitted if its entry distance is within the bound of this moment
                             * (the reference's test at its turn); a leaf goes to its test, an inner node's four entries are
                             * tested under the bound, the first the ray passes through is next, the others wait on the stack in
                             * the reference's order.  The stack's top is requested before the node, so that it is there when no
                             * entry of this step is next. */
                            const uint32_t ref = curRef;
                            const bool admitted = curEntry <= amax;
                            uint2 top = make_uint2(WIDE_NONE, 0u);
                            if (sp > 0)
                                top = pend[sp - 1];
                            curRef = WIDE_NONE;
                            if (admitted && ref < NODE_CHILD) {
                                leafPrim = ref;
                                state = S_LEAF;
                            } else if (admitted) {
                                const float4* w = sv.wideNodes + 8 * (size_t)(ref & NODE_INDEX_MASK);
                                const float4 lx = w[0], ly = w[1], lz = w[2], hx = w[3], hy = w[4], hz = w[5], rf = w[6];
                                const float amin = par.min_hit_distance;
                                /* entries 3, 2, 1, 0: the last one that passes (the first in the reference's order) stays in
                                 * registers, the one it displaces goes to the stack */
                                auto entryOf = [&](float lox, float loy, float loz, float hix, float hiy, float hiz, uint32_t r) {
                                    const float t0x = (lox - ps.o.x) * aux.inv.x, t0y = (loy - ps.o.y) * aux.inv.y, t0z = (loz - ps.o.z) * aux.inv.z;
                                    const float t1x = (hix - ps.o.x) * aux.inv.x, t1y = (hiy - ps.o.y) * aux.inv.y, t1z = (hiz - ps.o.z) * aux.inv.z;
                                    const float near = __builtin_fmaxf(__builtin_fmaxf(amin, __builtin_fminf(t0x, t1x)),
                                            __builtin_fmaxf(__builtin_fminf(t0y, t1y), __builtin_fminf(t0z, t1z)));
                                    const float far = __builtin_fminf(__builtin_fminf(amax, __builtin_fmaxf(t0x, t1x)),
                                            __builtin_fminf(__builtin_fmaxf(t0y, t1y), __builtin_fmaxf(t0z, t1z)));
                                    if (r != WIDE_NONE && near <= far) {
                                        if (curRef != WIDE_NONE)
                                            pend[sp++] = make_uint2(curRef, __float_as_uint(curEntry));
                                        curRef = r;
                                        curEntry = near;
                                    }
                                };
                                entryOf(lx.w, ly.w, lz.w, hx.w, hy.w, hz.w, __float_as_uint(rf.w));
                                entryOf(lx.z, ly.z, lz.z, hx.z, hy.z, hz.z, __float_as_uint(rf.z));
                                entryOf(lx.y, ly.y, lz.y, hx.y, hy.y, hz.y, __float_as_uint(rf.y));
                                entryOf(lx.x, ly.x, lz.x, hx.x, hy.x, hz.x, __float_as_uint(rf.x));
                            }
                            if (curRef == WIDE_NONE && top.x != WIDE_NONE) { /* nothing of this step is next: the stack's top is */
                                curRef = top.x;
                                curEntry = __uint_as_float(top.y);
                                sp--;
                            }
                            if (state == S_NODE && curRef == WIDE_NONE)
                                state = endOfRayState();
                        }
                        /* the few rays that walk the binary tree (a NaN slab distance is possible, or the bound has grown) */
                        if (__ballot(state == S_NODE && (aux.k & (RAY_MAY_NAN | RAY_WALK_BINARY))) != 0) {
                            if (state == S_NODE && (aux.k & (RAY_MAY_NAN | RAY_WALK_BINARY)))
                                binaryStep();
                        }
                    } else {
#pragma unroll
                        for (int step = 0; step < STEPS; step++) {
                            sec<COUNT>(lc, SEC_NODE_STEP, state == S_NODE);
                            if (state == S_NODE)
                                binaryStep();
                        }
                    }
                }
            }
        }
        if (COUNT && pick == S_NODE)
            sched[11] += (unsigned long long)(clock64() - tBlock);
        if (pick != S_NODE && (fused ? cShade > 0 : pick == S_SHADE)) {
            if (COUNT) {
                tBlock = clock64();
                sched[5]++;
                sched[6] += cShade;
            }
            if (state == S_SHADE) /* tracePath, one path component (wurblpt.hpp:131-252) */
                afterBlock(blockShade<F, COUNT>(sv, par, tri4, ps, best, lc, cTrav != 0 ? (int)args.waitBelow : 0));
            if (COUNT)
                sched[12] += (unsigned long long)(clock64() - tBlock);
        }
        if (pick != S_NODE && (fused ? cNee > 0 : pick == S_NEEEND)) {
            if (COUNT) {
                tBlock = clock64();
                sched[7]++;
                sched[8] += cNee;
            }
            sec<COUNT>(lc, SEC_NEE_END, state == S_NEEEND);
            sec<COUNT>(lc, SEC_NEE_END_LIGHT, state == S_NEEEND && ps.rayKind == RAY_NEE_LIGHT && best.prim == ps.getW(SLOT_NEE));
            if (state == S_NEEEND) /* the next-event ray's contribution, then the path continues */
                afterBlock(blockNeeEnd<F>(sv, par, tri4, ps, best));
            if (COUNT)
                sched[13] += (unsigned long long)(clock64() - tBlock);
        }
        /* last: in a fused round it also serves the lanes whose path has just ended above */
        if (pick != S_NODE && (fused ? __ballot(state == S_NEW) != 0 : pick == S_NEW)) {
            if (COUNT) {
                tBlock = clock64();
                sched[9]++;
                sched[10] += __popcll(__ballot(state == S_NEW));
            }
            sec<COUNT>(lc, SEC_NEW_SAMPLE, state == S_NEW);
            if (state == S_NEW) { /* the pixel's next sample (wurblpt.hpp:348-360), or nothing more */
                const bool passEnds = firstPass && (ps.getW(SLOT_ACC) >> 16) >= args.rowStop;
                const int next = passEnds ? (int)NEXT_DONE : blockNew<F>(fa, ps, sv);
                sec<COUNT>(lc, SEC_PIXEL_DONE, next == NEXT_DONE);
                if (next == NEXT_DONE) {
                    const uint32_t pxy = ps.getW(SLOT_SRDIR);
                    const size_t at = (size_t)(pxy >> 16) * args.width + (pxy & 0xffffu);
                    if (firstPass) {
                        args.carry[2 * at] = ps.base[SLOT_PRNG * WG];
                        args.carry[2 * at + 1] = ps.base[SLOT_ACC * WG];
                        args.cost[at] = (uint32_t)clock64() - args.cost[at];
                    } else {
                        /* SensorRGB::finishPixel (sensor_rgb.hpp:82-87) */
                        const Slot acc = ps.get(SLOT_ACC);
                        float* out = args.frame + 3 * at;
                        out[0] = args.invSamples * acc.x;
                        out[1] = args.invSamples * acc.y;
                        out[2] = args.invSamples * acc.z;
                    }
                }
                afterBlock(next);
            }
            fromPool();
            if (COUNT) /* shader clock spent per kind of block: [11] traversal [12] shade [13] nee-end [14] new */
                sched[14] += (unsigned long long)(clock64() - tBlock);
        }
        if (pick != S_NODE) {
            sec<COUNT>(lc, SEC_BEGIN_RAY, state == S_START);
            if (state == S_START)
                beginRay();
        }
    }

    if (COUNT && args.counters && inBlock) {
        atomicAdd((unsigned long long*)&args.counters->samples, (unsigned long long)args.samplesSqrt * args.samplesSqrt);
        atomicAdd((unsigned long long*)&args.counters->rays, (unsigned long long)lc.rays);
        atomicAdd((unsigned long long*)&args.counters->node_visits, (unsigned long long)lc.nodes);
        atomicAdd((unsigned long long*)&args.counters->leaf_tests, (unsigned long long)lc.leaves);
        atomicAdd((unsigned long long*)&args.counters->pdf_tests, (unsigned long long)lc.pdfs);
        atomicAdd((unsigned long long*)&args.counters->scatters, (unsigned long long)lc.scatters);
    }
    if (COUNT && args.schedStats && (threadIdx.x & 63) == 0) {
        for (int i = 0; i < 16; i++)
            atomicAdd(args.schedStats + i, sched[i]);
    }
    if (COUNT && args.schedStats && inBlock) { /* [16..23]: lane-weighted clock per section of the SHADE block */
        for (int i = 0; i < 8; i++)
            atomicAdd(args.schedStats + 16 + i, lc.shadeClock[i]);
    }
    if (COUNT && args.schedStats) { /* [24 ..]: executions of each stretch of code by waves, then by lanes (SEC_COUNT each) */
        for (int i = 0; i < SEC_COUNT; i++) {
            if (lc.secWave[i])
                atomicAdd(args.schedStats + 24 + i, (unsigned long long)lc.secWave[i]);
            if (lc.secLane[i])
                atomicAdd(args.schedStats + 24 + SEC_COUNT + i, (unsigned long long)lc.secLane[i]);
        }
    }
}

/* Launch of a product kernel.  With a pixel pool in the arguments and more workgroups than the device holds at once, the
 * launch shrinks to the resident workgroups and the counter starts behind their lanes; otherwise the pool stays unused. */
template<class Kernel>
inline void launchMaybePooled(Kernel kernel, const KernelArgs& args, dim3 grid, size_t ldsBytes, hipStream_t stream)
{
    KernelArgs a = args;
    if (a.pool) {
        int perCu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kernel, (int)WG, ldsBytes) != hipSuccess)
            perCu = 0;
        const uint64_t resident = (uint64_t)(perCu < 0 ? 0 : perCu) * a.cuCount;
        if (resident == 0 || grid.x <= resident
                || hipMemsetD32Async((hipDeviceptr_t)a.pool, (int)(resident * WG), 1, stream) != hipSuccess)
            a.pool = nullptr;
        else
            grid.x = (uint32_t)resident;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(WG), ldsBytes, stream, a);
}

constexpr uint32_t FEAT_BASIC = FEAT_GGX | FEAT_GLASS;
constexpr uint32_t FEAT_ALL = FEAT_TEXTURES | FEAT_MODPHONG | FEAT_ENVMAP | FEAT_LENS | FEAT_TWOSIDED | FEAT_GGX | FEAT_GLASS | FEAT_SPHERES;

/* getGroundTruth (wpt_k_groundtruth.hip): array[k] is the device array of GroundTruth bit k or NULL */
struct GroundTruthArgs {
    SceneView scene;
    wpt_camera cam, camPrev, camNext;
    wpt_params par;
    float t0, tPrev, tNext;
    uint32_t width, height;
    void* array[WPT_GT_ARRAY_COUNT];
};
void launchGroundTruth(const GroundTruthArgs& args, hipStream_t stream);

/* one launcher per instantiation, each defined in its own translation unit; sceneLdsBytes is the size of the scene
 * copy behind the cold path words in LDS (0 for the kernels that fetch the scene from HBM) */
/* wpt_k_order.hip: from the first pass's times to the second pass's order of pixels, longest first.  `work` is
 * 3 * ORDER_BUCKETS + 1 words of device memory; work[3 * ORDER_BUCKETS] receives the number of pixels in `order`. */
constexpr uint32_t ORDER_BUCKETS = 128;
void launchOrderBuild(const KernelArgs& args, uint32_t* order, uint32_t* work, hipStream_t stream);
void launchBasicLds(const KernelArgs& args, dim3 grid, size_t sceneLdsBytes, hipStream_t stream);
void launchBasic(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchBasicCount(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFull(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFullWide(const KernelArgs& args, dim3 grid, hipStream_t stream); /* the wide walk (SceneView::wideNodes) */
void launchFullCount(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFullRgl(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFullRglWide(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFullRglCount(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFullAnim(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFullAnimCount(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFullRglAnim(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFullRglAnimCount(const KernelArgs& args, dim3 grid, hipStream_t stream);

} /* namespace wptk */

#endif
