/*
 * wpt_wavefront_host.hip -- host side of the wavefront form (wpt_wavefront.inc.h): buffers, the groups' streams, the
 * iteration loop.  An iteration of a group is two launches on the group's stream, trace then shade; how many
 * iterations a frame takes is the largest number of rays one of its pixels traces, which only the device knows, so the
 * host enqueues iterations in batches and reads the number of pixels still queued behind each batch -- one batch
 * ahead, so that the device never waits for the host (the batches behind the last pixel are empty launches).
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "wpt_wavefront.inc.h"

namespace wptk {

namespace {

constexpr uint32_t BATCH = 32;        /* iterations between two looks at a group's queue length; divides WF_RING */
constexpr uint32_t MAX_GROUPS = 8;

/* what one render call needs besides its buffers: the groups' streams, events, and the pinned words the queue lengths are
 * copied to.  A call takes a set from the device's free list (or makes one) and gives it back: as many sets exist as
 * calls have ever run at once on the device, whatever threads made them (MPICoordinator starts its workers anew for every
 * frame), and none is destroyed -- at process exit the runtime may be gone before a destructor of ours would run. */
struct CallResources {
    hipStream_t stream[MAX_GROUPS];
    hipEvent_t batchDone[MAX_GROUPS][2];
    hipEvent_t fork, join[MAX_GROUPS];
    uint32_t* pinned; /* [MAX_GROUPS][2] */
    int device;
};

std::mutex g_resourcesLock;
std::vector<CallResources*> g_freeResources;

CallResources* acquireResources(int device)
{
    {
        std::lock_guard<std::mutex> lock(g_resourcesLock);
        for (size_t i = 0; i < g_freeResources.size(); i++) {
            if (g_freeResources[i]->device == device) {
                CallResources* r = g_freeResources[i];
                g_freeResources.erase(g_freeResources.begin() + long(i));
                return r;
            }
        }
    }
    CallResources* r = new CallResources;
    r->device = device;
    r->pinned = nullptr;
    bool ok = true;
    for (uint32_t g = 0; g < MAX_GROUPS && ok; g++) {
        ok = ok && hipStreamCreateWithFlags(&r->stream[g], hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&r->batchDone[g][0], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&r->batchDone[g][1], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&r->join[g], hipEventDisableTiming) == hipSuccess;
    }
    ok = ok && hipEventCreateWithFlags(&r->fork, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&r->pinned), MAX_GROUPS * 2 * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
    if (!ok) { /* whatever was made stays unused: this path means the device is out of resources anyway */
        (void)hipGetLastError();
        delete r;
        return nullptr;
    }
    return r;
}

void releaseResources(CallResources* r)
{
    std::lock_guard<std::mutex> lock(g_resourcesLock);
    g_freeResources.push_back(r);
}

struct Group {
    WfArgs args;
    uint32_t known;     /* upper bound of the pixels still queued (queues only shrink) */
    uint32_t iteration; /* next iteration to enqueue */
    bool active;
};

} /* namespace */

hipError_t renderWavefront(const KernelArgs& args, const WfLaunchers& kernels, const WfConfig& cfg, hipStream_t stream, uint32_t* launches)
{
#define WF_TRY(expr)                  \
    do {                              \
        const hipError_t e_ = (expr); \
        if (e_ != hipSuccess)         \
            return e_;                \
    } while (0)
    uint32_t launched = 0;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess)
        return hipErrorInvalidDevice;
    CallResources* tr = acquireResources(device);
    if (!tr)
        return hipErrorOutOfMemory;
    const uint32_t lanes = args.blockSize;
    /* groups of whole workgroups, none smaller than a quarter of what the device holds at once */
    const uint32_t resident = std::max(1u, args.cuCount) * WF_TRACE_WAVES * WG;
    /* one group by default: since a hit's light ray travels beside the continuation (half the iterations, fuller launches) the
     * overlap of two groups' kernels no longer pays for the smaller launches (measured BRDFs, 16 spp: 135.7 Msamples/s with one
     * group, 133.6 with two, 130.1 with three, 121.3 with four; round 3's form: 317 / 322 / 311 / 429 ms on the Sponza-class frame) */
    uint32_t groupCount = cfg.groups ? cfg.groups : 1u;
    groupCount = std::min(groupCount, MAX_GROUPS);
    while (groupCount > 1 && lanes / groupCount < resident / 4u)
        groupCount--;
    const uint32_t perGroup = ((lanes + groupCount - 1) / groupCount + WG - 1) / WG * WG;
    const uint32_t chunk = cfg.chunk ? cfg.chunk : 128u;
    /* the top of the tree in LDS: with the 8 KiB of staging, four workgroups per compute unit hold 32 KiB each */
    const uint32_t topNodes = cfg.topNodes == 0xffffffffu ? 0u : std::min(std::min(cfg.topNodes ? cfg.topNodes : 768u, 768u), args.sv.nodeCount);
    int perCu = wfTraceBlocksPerCu(kernels.spheres, size_t(topNodes) * 32);
    perCu = perCu < 1 ? 1 : (perCu > WF_TRACE_WAVES ? WF_TRACE_WAVES : perCu);
    if (cfg.tracePerCu)
        perCu = std::max(1, std::min(perCu, int(cfg.tracePerCu)));
    const uint32_t traceResident = uint32_t(perCu) * std::max(1u, args.cuCount);
    const bool perKind = cfg.shadePerKind != 0;

    float4* state = nullptr;
    uint32_t* queues = nullptr;
    WfIter* rings = nullptr;
    std::vector<Group> groups;
    const size_t queueWords = size_t(perGroup) * (2 + WF_BUCKETS);
    auto body = [&]() -> hipError_t {
    WF_TRY(hipMallocAsync(reinterpret_cast<void**>(&state), size_t(lanes) * WF_SLOTS * sizeof(float4), stream));
    WF_TRY(hipMallocAsync(reinterpret_cast<void**>(&queues), queueWords * groupCount * sizeof(uint32_t), stream));
    WF_TRY(hipMallocAsync(reinterpret_cast<void**>(&rings), size_t(groupCount) * WF_RING * sizeof(WfIter), stream));
    WF_TRY(hipMemsetAsync(rings, 0, size_t(groupCount) * WF_RING * sizeof(WfIter), stream));
    WF_TRY(hipEventRecord(tr->fork, stream));
    for (uint32_t g = 0; g < groupCount; g++) {
        Group gr;
        gr.args.k = args;
        gr.args.k.pool = nullptr;
        gr.args.state = state;
        uint32_t* q = queues + queueWords * g;
        gr.args.rayQueue[0] = q;
        gr.args.rayQueue[1] = q + perGroup;
        gr.args.bucketQueue = q + 2 * size_t(perGroup);
        gr.args.ring = rings + size_t(g) * WF_RING;
        gr.args.laneFirst = g * perGroup;
        gr.args.laneCount = gr.args.laneFirst < lanes ? std::min(perGroup, lanes - gr.args.laneFirst) : 0u;
        gr.args.iteration = 0;
        gr.args.buckets = cfg.buckets;
        gr.args.chunk = chunk;
        gr.args.refillIdle = cfg.refillIdle ? cfg.refillIdle : 16u;
        gr.args.leafBias = cfg.leafBias ? cfg.leafBias : 32u;
        gr.args.kindMask = (1u << WF_BUCKETS) - 1u;
        gr.args.stepBudget = cfg.stepBudget == 0xffffffffu ? 0u : (cfg.stepBudget ? cfg.stepBudget : 512u);
        gr.args.topNodes = topNodes;
        gr.known = gr.args.laneCount;
        gr.iteration = 0;
        gr.active = gr.args.laneCount != 0;
        groups.push_back(gr);
        WF_TRY(hipStreamWaitEvent(tr->stream[g], tr->fork, 0));
        if (gr.active) {
            /* the first samples' camera rays; they are queued as iteration 0's rays */
            groups[g].args.iteration = ~0u; /* appends to iteration 0's queue */
            kernels.init(groups[g].args, dim3((gr.args.laneCount + WG - 1) / WG), tr->stream[g]);
            launched++;
        }
    }
    {
        auto enqueueBatch = [&](uint32_t g, uint32_t parity) -> hipError_t {
            Group& gr = groups[g];
            hipStream_t s = tr->stream[g];
            /* counters of the iterations this batch's shades append to: entries iteration + 1 .. iteration + BATCH of the
             * ring (the batch's first entry holds the queue length the batch before left there) */
            if (gr.iteration != 0) { /* the first batch finds the whole ring cleared */
                const uint32_t from = (gr.iteration + 1) & (WF_RING - 1);
                const uint32_t firstPiece = std::min(BATCH, WF_RING - from);
                hipError_t e = hipMemsetAsync(gr.args.ring + from, 0, firstPiece * sizeof(WfIter), s);
                if (e == hipSuccess && firstPiece < BATCH)
                    e = hipMemsetAsync(gr.args.ring, 0, (BATCH - firstPiece) * sizeof(WfIter), s);
                if (e != hipSuccess)
                    return e;
            }
            const uint32_t traceGroups = std::max(1u, std::min(traceResident, (gr.known + chunk * (WG / 64) - 1) / (chunk * (WG / 64))));
            const uint32_t shadeGroups = (gr.known + WG - 1) / WG + (gr.args.buckets ? WF_BUCKETS : 0u);
            for (uint32_t i = 0; i < BATCH; i++) {
                gr.args.iteration = gr.iteration + i;
                launchWfTrace(kernels.spheres, gr.args, dim3(traceGroups), s);
                if (perKind && gr.args.buckets) {
                    /* measurements: one shade launch per kind, so that a kernel trace tells what each kind costs */
                    for (uint32_t kd = 0; kd < WF_BUCKETS; kd++) {
                        gr.args.kindMask = 1u << kd;
                        kernels.shade(gr.args, dim3(shadeGroups), s);
                    }
                    gr.args.kindMask = (1u << WF_BUCKETS) - 1u;
                    launched += WF_BUCKETS - 1;
                } else {
                    kernels.shade(gr.args, dim3(shadeGroups), s);
                }
            }
            launched += 2 * BATCH;
            gr.iteration += BATCH;
            hipError_t e = hipMemcpyAsync(tr->pinned + 2 * g + parity, &gr.args.ring[gr.iteration & (WF_RING - 1)].rayCount, sizeof(uint32_t),
                    hipMemcpyDeviceToHost, s);
            if (e == hipSuccess)
                e = hipEventRecord(tr->batchDone[g][parity], s);
            return e;
        };
        uint32_t parity = 0;
        for (uint32_t g = 0; g < groupCount; g++)
            if (groups[g].active)
                WF_TRY(enqueueBatch(g, parity));
        for (;;) {
            /* one batch ahead: enqueue the next before waiting for the one in flight */
            for (uint32_t g = 0; g < groupCount; g++)
                if (groups[g].active)
                    WF_TRY(enqueueBatch(g, parity ^ 1u));
            bool any = false;
            for (uint32_t g = 0; g < groupCount; g++) {
                if (!groups[g].active)
                    continue;
                WF_TRY(hipEventSynchronize(tr->batchDone[g][parity]));
                groups[g].known = tr->pinned[2 * g + parity];
                if (groups[g].known == 0)
                    groups[g].active = false; /* the batch already enqueued behind it finds empty queues */
                else
                    any = true;
            }
            parity ^= 1u;
            if (!any)
                break;
        }
    }
    return hipGetLastError();
    };
    const hipError_t status = body();
    /* the caller's stream goes on behind the groups' streams; the buffers are freed behind that */
    for (uint32_t g = 0; g < groupCount && g < MAX_GROUPS; g++) {
        if (hipEventRecord(tr->join[g], tr->stream[g]) == hipSuccess)
            (void)hipStreamWaitEvent(stream, tr->join[g], 0);
        if (status != hipSuccess)
            (void)hipStreamSynchronize(tr->stream[g]);
    }
    for (void* p : { static_cast<void*>(state), static_cast<void*>(queues), static_cast<void*>(rings) })
        if (p)
            (void)hipFreeAsync(p, stream);
    if (launches)
        *launches = launched;
    /* everything of this call on the groups' streams is done or (after an error) waited for: the set may serve another call */
    for (uint32_t g = 0; g < groupCount && g < MAX_GROUPS; g++)
        (void)hipStreamSynchronize(tr->stream[g]);
    releaseResources(tr);
    return status;
#undef WF_TRY
}

} /* namespace wptk */
