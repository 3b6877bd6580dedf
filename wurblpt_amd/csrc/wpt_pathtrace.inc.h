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
 *  - BVH traversal is stackless.  The reference walks its depth-first node array with a stack,
 *    left child first (bvh.hpp:277-311); in that layout the node a pop returns to is always
 *    the first node after the current subtree, so each device node carries that index
 *    ("skip") and the walk is: box hit & inner -> next node; otherwise -> skip.  Same visiting
 *    order, no stack, no LDS or scratch traffic for it.
 *
 *  - Small scenes (nodes + triangle positions up to 48 KiB, e.g. the Cornell box: 4 KiB) are
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
constexpr uint32_t NODE_INNER = 0xffffffffu; /* device node: marker in the primitive slot */
constexpr uint32_t NODE_EMPTY = 0xfffffffeu;
constexpr uint32_t LDS_SCENE_MAX_BYTES = 48 * 1024;
/* Near-first walk over the octant copies of the BVH (template parameter ORDERED).  Measured on
 * the Sponza-class scene: only 10 % fewer node visits (1112 vs 1243 per sample; the reference's
 * tree is 50 levels deep, so most visits are the two boxes per level on the way down) for 4 more
 * registers per lane, which costs more than it saves.  Kept as an option, switched off. */
constexpr bool ORDERED_KERNELS = false;

struct KernelArgs {
    SceneView sv;
    wpt_camera cam;
    wpt_params par;
    uint32_t width, height, samplesSqrt;
    uint32_t blockStart, blockSize;
    uint32_t tiled; /* 1: a wave covers an 8x8 pixel tile (block is whole rows, multiple of 8) */
    /* bands (product kernel only): with bandStride > 0 the launch covers the bands of bandPixels consecutive pixels
     * whose index is bandFirst, bandFirst + bandStride, ... -- one rank's interleaved share of the frame in one launch;
     * blockStart is 0 and blockSize the number of lanes (pixels behind the end of the frame stay idle) */
    uint32_t bandPixels, bandFirst, bandStride;
    uint32_t leaveEighths; /* scheduler: leave the NODE loop when fewer than this many eighths of the entering lanes remain */
    uint32_t heavyMin;     /* scheduler: lanes a long block needs before it runs */
    uint32_t leafBias;     /* scheduler: leaf tests run when waiting lanes * leafBias >= walking lanes * 8 */
    uint32_t patience;     /* ray-pool kernel: traversal rounds a wave may spend before it shades what it has */
    uint32_t fuse;         /* scheduler: 1 = one long round serves SHADE, NEE-END and NEW lanes together */
    uint32_t travWaves;    /* LDS-state kernel: waves of a workgroup that hold traversal contexts */
    uint32_t heavyWaves;   /* LDS-state kernel: waves of a workgroup that run heavy batches */
    uint32_t* pixelCounter; /* LDS-state kernel: next unassigned pixel of the block (zero at launch) */
    float4* wfState;        /* state-in-memory kernel: 12 float4 per slot, wfSlots slots per workgroup */
    uint32_t wfSlots;
    float* frame;
    wpt_counters* counters;
    unsigned long long* schedStats; /* COUNT builds: 16 scheduler statistics, or NULL */
    uint32_t* status;               /* set to 1 by a launch that had to abort (bounded waits) */
};

/* lane states, in scheduling priority order for ties */
enum { S_NODE = 0, S_LEAF = 1, S_SHADE = 2, S_NEEEND = 3, S_NEW = 4, S_DONE = 5 };

template<uint32_t F, bool COUNT, bool LDSSCENE, int OCC, bool ORDERED, bool PAIRS = false>
__global__ __launch_bounds__(WG, OCC) void wpt_pathtrace(const KernelArgs args)
{
    extern __shared__ float4 ldsScene[];

    const SceneView& sv = args.sv;
    const wpt_params& par = args.par;
    const uint32_t nodeCount = sv.nodeCount;

    if (LDSSCENE) {
        /* nodes (2 x float4 each) followed by the triangle positions (3 x float4 each) */
        const uint32_t n4 = 2 * nodeCount * (ORDERED ? 8u : 1u), t4 = 3 * sv.triCount;
        for (uint32_t i = threadIdx.x; i < n4; i += WG)
            ldsScene[i] = sv.nodes[i];
        for (uint32_t i = threadIdx.x; i < t4; i += WG)
            ldsScene[n4 + i] = sv.triGeom[i];
        __syncthreads();
    }
    auto node4 = [&](uint32_t i) -> float4 {
        if constexpr (LDSSCENE)
            return ldsScene[i];
        else
            return sv.nodes[i];
    };
    auto tri4 = [&](uint32_t i) -> float4 {
        if constexpr (LDSSCENE)
            return ldsScene[2 * nodeCount * (ORDERED ? 8u : 1u) + i];
        else
            return sv.triGeom[i];
    };

    /* lane -> pixel */
    const uint32_t gid = blockIdx.x * WG + threadIdx.x;
    bool inBlock = gid < args.blockSize;
    uint32_t pixel;
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
    const uint32_t samples = args.samplesSqrt * args.samplesSqrt;
    FrameArgs fa;
    fa.cam = args.cam;
    fa.par = args.par;
    fa.width = args.width;
    fa.height = args.height;
    fa.samplesSqrt = args.samplesSqrt;

    /* ---- per-lane state: the pixel's path (wpt_blocks.h) and the traversal registers ---- */
    PathState ps;
    pathStateInit(ps, pixel, args.width);
    LaneCounters lc = { 0, 0, 0, 0, 0, { 0, 0, 0, 0, 0, 0, 0, 0 } };
    /* wave-level scheduler statistics (COUNT builds): rounds and lane counts per state */
    unsigned long long sched[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    int state = inBlock ? S_NEW : S_DONE;
    RayAux aux = rayAux(ps.ray.d);
    uint32_t node = 0, leafPrim = 0;
    uint32_t nodeBase = 0; /* ORDERED: first node of the ray's octant copy of the BVH */
    bool exact = !ORDERED; /* true: the reference's own walk (copy 0, exact bounds) */
    float amax = k_maxval;
    float amaxCull = k_maxval, second = k_maxval; /* ORDERED: widened bound, 2nd closest candidate */
    Candidate best;
    best.prim = NO_HIT;
    best.a = best.invDet = best.U = best.V = best.W = 0.0f;

    /* start the traversal of ps.ray */
    auto beginRay = [&]() {
        aux = rayAux(ps.ray.d);
        node = 0;
        amax = k_maxval;
        best.prim = NO_HIT;
        if (ORDERED) {
            const uint32_t octant = (ps.ray.d.x < 0.0f ? 1u : 0u) | (ps.ray.d.y < 0.0f ? 2u : 0u) | (ps.ray.d.z < 0.0f ? 4u : 0u);
            nodeBase = octant * nodeCount;
            exact = false;
            amaxCull = k_maxval;
            second = k_maxval;
        }
        state = S_NODE;
        if (COUNT)
            lc.rays++;
    };
    /* State after the walk has left the tree.  ORDERED: the near-first walk found the closest
     * candidate `best` and the runner-up distance `second`; if no other candidate lies within
     * (1 + 2^-13) of it, the reference's unordered walk must end with the same candidate (it
     * accepts it whenever it reaches it, and nothing can replace it).  Otherwise candidates
     * tie and the winner depends on the reference's visiting order, so the ray walks again in
     * exactly that order (copy 0, exact bounds) starting from amax = best.a * (1 + 2^-13),
     * which gives the reference's result: everything farther is superseded in its walk anyway. */
    auto endOfRayState = [&]() {
        if (ORDERED && !exact && best.prim != NO_HIT && !(second > best.a * 1.0001220703125f)) {
            exact = true;
            nodeBase = 0;
            node = 0;
            amax = best.a * 1.0001220703125f;
            best.prim = NO_HIT;
            return (int)S_NODE;
        }
        return ps.rayKind == RAY_PATH ? (int)S_SHADE : (int)S_NEEEND;
    };
    /* what a block of wpt_blocks.h asks for next */
    auto afterBlock = [&](int next) {
        if (next == NEXT_TRACE)
            beginRay();
        else
            state = next == NEXT_NEW ? (int)S_NEW : (int)S_DONE;
    };

    for (;;) {
        /* ---- the wave's scheduler ----
         * Traversal (NODE steps and LEAF tests) is one block with its own inner policy; the long
         * blocks (SHADE, NEE-END, NEW) run when they are well filled, or when no traversal work
         * is left in the wave.  Waiting lanes lose nothing but time: every lane still executes
         * its own operations in order. */
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
            for (;;) {
                const int nNode = __popcll(__ballot(state == S_NODE));
                const int nLeaf = __popcll(__ballot(state == S_LEAF));
                if (nNode + nLeaf < leaveBelow)
                    break;
                /* a leaf test is ~3 node steps long; it runs once enough lanes wait for it
                 * (leafBias/8 of the walking lanes), because waiting lanes thin out the walk */
                if (nLeaf * (int)args.leafBias >= nNode * 8 && nLeaf > 0) {
                    if (COUNT) {
                        sched[3]++;
                        sched[4] += nLeaf;
                    }
                    if (state == S_LEAF) {
                        /* HitableTriangle::hit, candidate part (hitable_triangle.hpp:189-271) */
                        if (COUNT)
                            lc.leaves++;
                        Candidate c;
                        const float bound = (ORDERED && !exact) ? amaxCull : amax;
                        bool accepted;
                        if ((F & FEAT_SPHERES) && (leafPrim & PRIM_SPHERE)) {
                            /* HitableSphere::hit (hitable_sphere.hpp:104-147) */
                            c.invDet = c.U = c.V = c.W = 0.0f;
                            accepted = sphereTest(sphereNow<F>(sv, ps, sv.spheres[leafPrim & ~PRIM_SPHERE]), ps.ray.o, ps.ray.d, par.min_hit_distance, bound, c.a);
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
                            accepted = triangleTest(v0, v1, v2, ps.ray.o, aux, par.min_hit_distance, bound, c);
                        }
                        if (accepted) {
                            c.prim = leafPrim;
                            if (ORDERED && !exact) {
                                /* keep the closest; remember how close the runner-up came */
                                if (best.prim == NO_HIT || c.a < best.a) {
                                    second = best.prim != NO_HIT && best.a < second ? best.a : second;
                                    best = c;
                                    amax = c.a;
                                    amaxCull = amax * 1.000244140625f; /* 1 + 2^-12 */
                                } else {
                                    second = c.a < second ? c.a : second;
                                }
                            } else {
                                best = c;
                                amax = c.a;
                            }
                        }
                        node = node + 1; /* a leaf's subtree is the leaf itself */
                        state = node >= nodeCount ? endOfRayState() : S_NODE;
                    }
                } else {
                    if (COUNT) {
                        sched[1]++;
                        sched[2] += nNode;
                    }
                    if (state == S_NODE) {
                        /* AABB::mayHit + the stackless form of BVH::hit's walk; true when the walk
                         * descends into the node's first child, which is the next node in memory */
                        auto nodeStep = [&](const float4& n0, const float4& n1) {
                            if (COUNT)
                                lc.nodes++;
                            const uint32_t skip = __float_as_uint(n1.z);
                            const uint32_t prim = __float_as_uint(n1.w);
                            const bool hit = boxTest(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), ps.ray.o, aux.inv, par.min_hit_distance,
                                    (ORDERED && !exact) ? amaxCull : amax);
                            /* select form of: hit & inner -> next node; hit & leaf -> test it; else -> skip */
                            const bool toLeaf = hit && prim < NODE_EMPTY;
                            const bool down = hit && prim == NODE_INNER;
                            const uint32_t next = down ? node + 1 : skip;
                            leafPrim = toLeaf ? prim : leafPrim;
                            node = toLeaf ? node : next;
                            state = toLeaf ? (int)S_LEAF : (int)S_NODE;
                            if (!toLeaf && node >= nodeCount)
                                state = endOfRayState();
                            return down;
                        };
                        const uint32_t at = 2 * (nodeBase + node);
                        const float4 n0 = node4(at), n1 = node4(at + 1);
                        if (PAIRS) {
                            /* Option for scenes in HBM, measured slower and off (Sponza-class 0.95x,
                             * 10 M triangles 0.93x): a node step is a dependent memory round trip, so
                             * the neighbouring node (the first child, usually in the same 128-byte
                             * line) is fetched with it and a lane that descends takes two steps per trip */
                            const uint32_t at1 = node + 1 < nodeCount ? at + 2 : at;
                            const float4 m0 = node4(at1), m1 = node4(at1 + 1);
                            if (nodeStep(n0, n1))
                                nodeStep(m0, m1);
                        } else {
                            nodeStep(n0, n1);
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
                afterBlock(blockShade<F, COUNT>(sv, par, tri4, ps, best, lc));
            if (COUNT)
                sched[12] += (unsigned long long)(clock64() - tBlock);
        }
        if (pick != S_NODE && (fused ? cNee > 0 : pick == S_NEEEND)) {
            if (COUNT) {
                tBlock = clock64();
                sched[7]++;
                sched[8] += cNee;
            }
            if (state == S_NEEEND) /* the next-event ray's contribution, then the path continues */
                afterBlock(blockNeeEnd<F>(sv, par, ps, best));
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
            if (state == S_NEW) /* the pixel's next sample (wurblpt.hpp:348-360), or nothing more */
                afterBlock(blockNew<F>(fa, ps, &sv));
            if (COUNT) /* shader clock spent per kind of block: [11] traversal [12] shade [13] nee-end [14] new */
                sched[14] += (unsigned long long)(clock64() - tBlock);
        }
    }

    if (inBlock) {
        /* SensorRGB::finishPixel (sensor_rgb.hpp:82-87) */
        const float invSamples = 1.0f / (float)samples;
        float* out = args.frame + 3 * (size_t)pixel;
        out[0] = invSamples * ps.acc0;
        out[1] = invSamples * ps.acc1;
        out[2] = invSamples * ps.acc2;
    }
    if (COUNT && args.counters && inBlock) {
        atomicAdd((unsigned long long*)&args.counters->samples, (unsigned long long)samples);
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
}

constexpr uint32_t FEAT_BASIC = FEAT_GGX | FEAT_GLASS;
constexpr uint32_t FEAT_ALL = FEAT_TEXTURES | FEAT_MODPHONG | FEAT_ENVMAP | FEAT_LENS | FEAT_TWOSIDED | FEAT_GGX | FEAT_GLASS | FEAT_SPHERES;

/* one launcher per instantiation, each defined in its own translation unit;
 * ldsBytes is the dynamic LDS size (0 for the HBM variants) */
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

void launchBasicLds(const KernelArgs& args, dim3 grid, size_t ldsBytes, hipStream_t stream);
void launchBasicLdsPairs(const KernelArgs& args, dim3 grid, size_t ldsBytes, hipStream_t stream);
void launchBasic(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchBasicCount(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFull(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFullCount(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFullRgl(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFullRglCount(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFullAnim(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFullAnimCount(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFullRglAnim(const KernelArgs& args, dim3 grid, hipStream_t stream);
void launchFullRglAnimCount(const KernelArgs& args, dim3 grid, hipStream_t stream);

} /* namespace wptk */

#endif
