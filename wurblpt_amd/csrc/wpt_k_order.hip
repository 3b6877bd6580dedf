/* wpt_k_order.hip -- the order in which the second pass of a frame hands out its pixels: by the time the first pass spent
 * on them (quarter octaves of the shader clock), longest first.  The pixels that are still being rendered when the launch
 * runs out of pixels then are short ones.  Order within a class is whatever the atomics give: a pixel's value does not
 * depend on when or where it is rendered. */
#include "wpt_pathtrace.inc.h"

namespace wptk {

namespace {

constexpr uint32_t OB = 1024; /* threads per workgroup here */

/* class of a time: 4 * floor(log2) + the two bits below the leading one; larger = longer */
__device__ inline uint32_t costClass(uint32_t ticks)
{
    if (ticks < 8u)
        return ticks >> 1;
    const uint32_t msb = 31u - (uint32_t)__clz((int)ticks);
    const uint32_t c = (msb << 2) | ((ticks >> (msb - 2u)) & 3u);
    return c < ORDER_BUCKETS ? c : ORDER_BUCKETS - 1u;
}

/* The unit of the order is the wave's worth of 64 consecutive lane indices (an 8x8 tile where the launch is tiled):
 * neighbouring pixels take like paths, and a wave that renders neighbours diverges less than one that renders pixels
 * from all over the frame (measured: pixels ordered one by one cost the second pass 9 %).  Its time is the mean of
 * its pixels' times. */
__device__ inline uint32_t unitClass(bool have, uint32_t ticks, uint32_t& members, uint32_t& rankInUnit)
{
    const unsigned long long lanes = __ballot(have);
    members = (uint32_t)__popcll(lanes);
    rankInUnit = __builtin_amdgcn_mbcnt_hi((uint32_t)(lanes >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lanes, 0u));
    unsigned long long sum = have ? ticks : 0u;
    for (int d = 32; d >= 1; d >>= 1)
        sum += __shfl_xor(sum, d);
    return costClass(members ? (uint32_t)(sum / members) : 0u);
}

__global__ __launch_bounds__(OB) void orderHistogram(const KernelArgs args, uint32_t* work)
{
    __shared__ uint32_t local[ORDER_BUCKETS];
    if (threadIdx.x < ORDER_BUCKETS)
        local[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t gid = blockIdx.x * OB + threadIdx.x;
    uint32_t pixel, members, rankInUnit;
    const bool have = lanePixel(args, gid, pixel);
    const uint32_t c = unitClass(have, have ? args.cost[pixel] : 0u, members, rankInUnit);
    if (have && rankInUnit == 0)
        atomicAdd(&local[c], members);
    __syncthreads();
    if (threadIdx.x < ORDER_BUCKETS && local[threadIdx.x])
        atomicAdd(&work[threadIdx.x], local[threadIdx.x]);
}

/* work[0..B): counts -> work[B..2B): where each class starts, longest class first; work[3B]: pixels in all */
__global__ void orderOffsets(uint32_t* work)
{
    uint32_t at = 0;
    for (int c = (int)ORDER_BUCKETS - 1; c >= 0; c--) {
        work[ORDER_BUCKETS + c] = at;
        at += work[c];
    }
    work[3 * ORDER_BUCKETS] = at;
}

__global__ __launch_bounds__(OB) void orderScatter(const KernelArgs args, uint32_t* order, uint32_t* work)
{
    __shared__ uint32_t local[ORDER_BUCKETS], start[ORDER_BUCKETS];
    if (threadIdx.x < ORDER_BUCKETS)
        local[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t gid = blockIdx.x * OB + threadIdx.x;
    uint32_t pixel, members, rankInUnit;
    const bool have = lanePixel(args, gid, pixel);
    const uint32_t c = unitClass(have, have ? args.cost[pixel] : 0u, members, rankInUnit);
    uint32_t unitAt = 0;
    if (have && rankInUnit == 0)
        unitAt = atomicAdd(&local[c], members);
    unitAt = __shfl(unitAt, __ffsll((long long)__ballot(have && rankInUnit == 0)) - 1);
    __syncthreads();
    if (threadIdx.x < ORDER_BUCKETS && local[threadIdx.x])
        start[threadIdx.x] = atomicAdd(&work[ORDER_BUCKETS + threadIdx.x], local[threadIdx.x]);
    __syncthreads();
    if (have)
        order[start[c] + unitAt + rankInUnit] = pixel;
}

}

void launchOrderBuild(const KernelArgs& args, uint32_t* order, uint32_t* work, hipStream_t stream)
{
    (void)hipMemsetAsync(work, 0, (3 * ORDER_BUCKETS + 1) * sizeof(uint32_t), stream);
    const dim3 grid((args.blockSize + OB - 1) / OB);
    hipLaunchKernelGGL(orderHistogram, grid, dim3(OB), 0, stream, args, work);
    hipLaunchKernelGGL(orderOffsets, dim3(1), dim3(1), 0, stream, work);
    hipLaunchKernelGGL(orderScatter, grid, dim3(OB), 0, stream, args, order, work);
}

}
