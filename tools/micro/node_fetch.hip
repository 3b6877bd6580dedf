/* node_fetch.hip -- what bounds dependent random fetches of 32-byte BVH nodes on MI355X?
 * Every lane walks its own chain through an array of 32-byte records: next index = hash of what it loaded.
 *   variant 0: a lane loads its record as two dwordx4 (what the path tracer does)
 *   variant 1: a pair of lanes loads each of its two records together (lane 2i: first half, lane 2i+1: second half; two
 *              passes), halves exchanged by DPP: half the distinct lines per instruction
 *   variant 2: a lane loads the 64-byte aligned pair of records that holds its record (four dwordx4)
 *   variant 3: a lane loads only the first 16 bytes
 *   variant 4: as 0, but the chain moves to the adjacent record (index + 1) three times out of four
 * build: hipcc --offload-arch=gfx950 -O3 node_fetch.hip -o node_fetch;  run: ./node_fetch <records> <steps> */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template<int VARIANT, int WAVES>
__global__ __launch_bounds__(256, WAVES) void walk(const uint4* __restrict__ rec, uint32_t n, uint32_t steps, uint32_t* out)
{
    uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x + 1u) % n;
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; s++) {
        uint4 a, b;
        if (VARIANT == 0 || VARIANT == 4) {
            a = rec[2 * (size_t)idx];
            b = rec[2 * (size_t)idx + 1];
        } else if (VARIANT == 1) {
            const uint32_t odd = threadIdx.x & 1u;
            /* pass A: the even lane's record; pass B: the odd lane's */
            const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, 0xb1, 0xf, 0xf, false); /* quad_perm [1,0,3,2] */
            const uint32_t idxA = odd ? other : idx, idxB = odd ? idx : other;
            const uint4 qa = rec[2 * (size_t)idxA + odd];
            const uint4 qb = rec[2 * (size_t)idxB + odd];
            /* even lane: a = qa (own first half), b = partner's qa (own second half); odd lane: a = partner's qb, b = qb */
            const uint4 mineFirst = odd ? qb : qa, send = odd ? qa : qb;
            uint4 got;
            got.x = (uint32_t)__builtin_amdgcn_mov_dpp((int)send.x, 0xb1, 0xf, 0xf, false);
            got.y = (uint32_t)__builtin_amdgcn_mov_dpp((int)send.y, 0xb1, 0xf, 0xf, false);
            got.z = (uint32_t)__builtin_amdgcn_mov_dpp((int)send.z, 0xb1, 0xf, 0xf, false);
            got.w = (uint32_t)__builtin_amdgcn_mov_dpp((int)send.w, 0xb1, 0xf, 0xf, false);
            a = odd ? got : mineFirst;
            b = odd ? mineFirst : got;
        } else if (VARIANT == 2) {
            const size_t base = 4 * (size_t)(idx >> 1);
            const uint4 q0 = rec[base], q1 = rec[base + 1], q2 = rec[base + 2], q3 = rec[base + 3];
            a = (idx & 1u) ? q2 : q0;
            b = (idx & 1u) ? q3 : q1;
            acc += q0.y ^ q3.z;
        } else {
            a = rec[2 * (size_t)idx];
            b = a;
        }
        acc += a.y + b.z;
        const uint32_t h = mix(idx ^ a.x ^ b.w ^ s);
        if (VARIANT == 4 && (h & 3u) != 0u)
            idx = idx + 1 < n ? idx + 1 : 0;
        else
            idx = h % n;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc + idx;
}

template<int VARIANT, int WAVES> double run(const uint4* rec, uint32_t n, uint32_t steps, uint32_t* out, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((walk<VARIANT, WAVES>), dim3(blocks), dim3(256), 0, 0, rec, n, 16u, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((walk<VARIANT, WAVES>), dim3(blocks), dim3(256), 0, 0, rec, n, steps, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * 256.0 * steps / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 535887u;
    const uint32_t steps = argc > 2 ? (uint32_t)atol(argv[2]) : 2000u;
    std::vector<uint32_t> host((size_t)n * 8);
    uint32_t x = 12345;
    for (auto& v : host) { x = x * 1664525u + 1013904223u; v = x; }
    uint4* rec; uint32_t* out;
    hipMalloc(&rec, (size_t)n * 32);
    hipMalloc(&out, 4096 * 256 * 4);
    hipMemcpy(rec, host.data(), (size_t)n * 32, hipMemcpyHostToDevice);
    printf("%u records of 32 B (%.1f MB), %u steps per lane; G record fetches per second\n", n, n * 32.0 / 1e6, steps);
    printf("%-44s %10s %10s %10s %10s\n", "variant", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD", "8w, 1/2 grid");
#define ROW(V, NAME) printf("%-44s %10.1f %10.1f %10.1f %10.1f\n", NAME, run<V, 2>(rec, n, steps, out, 512), run<V, 4>(rec, n, steps, out, 1024), run<V, 8>(rec, n, steps, out, 2048), run<V, 8>(rec, n, steps, out, 1024));
    ROW(0, "0: two dwordx4 per lane");
    ROW(1, "1: lane pairs share a record, DPP exchange");
    ROW(2, "2: aligned 64-byte pair, four dwordx4");
    ROW(3, "3: first 16 bytes only");
    ROW(4, "4: as 0, 3 of 4 steps to the next record");
    return 0;
}
