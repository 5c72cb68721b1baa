// membench_copyorder.hip -- does the ORDER in which the eight XCDs walk a flat copy matter?  (round 5, after the 1-D tile order changed, R5.9)
// One 16-byte vector per thread, nontemporal, 256 threads per block = 4 KiB per block; block b is remapped as in sg1d_tile_body:
// s < 0: launch order; s == 0: every XCD one contiguous eighth; s > 0: chunks of 2^s blocks dealt to the XCDs round robin.
// Also WIDE blocks: each thread moves V vectors 4 KiB apart inside a 4 V KiB block (V = 2, 4, 8) -- the 1-D kernel's 8 KiB-per-wave tiles.
//   hipcc --offload-arch=gfx950 -O3 -o tools/membench_copyorder tools/membench_copyorder.hip && tools/membench_copyorder
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int V>
__global__ __launch_bounds__(256) void k_copy(const u4 *__restrict__ in, u4 *__restrict__ out, unsigned nblocks, int s)
{
    const unsigned nb8 = nblocks >> 3;
    unsigned blk = blockIdx.x;
    if (s >= 0 && blk < nb8 * 8u) {
        if (s == 0) blk = (blk & 7u) * nb8 + (blk >> 3);
        else {
            const unsigned span = 8u << s, q = blk >> (s + 3);
            if ((q + 1u) * span <= nb8 * 8u) { const unsigned r = blk & (span - 1u); blk = (((q << 3) + (r & 7u)) << s) + (r >> 3); }
        }
    }
    const size_t base = (size_t)blk * 256 * V + threadIdx.x;
    u4 v[V];
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = __builtin_nontemporal_load(in + base + 256 * i);
#pragma unroll
    for (int i = 0; i < V; ++i) __builtin_nontemporal_store(v[i], out + base + 256 * i);
}

template <int V>
static float run(const u4 *in, u4 *out, size_t nvec, int s)
{
    const unsigned nblocks = (unsigned)(nvec / (256 * V));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_copy<V>, dim3(nblocks), dim3(256), 0, 0, in, out, nblocks, s);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_copy<V>, dim3(nblocks), dim3(256), 0, 0, in, out, nblocks, s); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[2];
}

int main()
{
    const size_t bytes = (size_t)16 << 30, nvec = bytes / 16;
    const int orders[] = {-1, 0, 2, 4, 6, 8, 10, 12};
    std::vector<void *> keep;
    for (int alloc = 0; alloc < 3; ++alloc) {
        u4 *in, *out; CK(hipMalloc((void **)&in, bytes)); CK(hipMalloc((void **)&out, bytes)); keep.push_back(in); keep.push_back(out);
        CK(hipMemset(in, 1, bytes));
        printf("## buffer pair %d (16 GiB each): fraction of 8 TB/s by block order (launch, eighths, chunks of 2^s blocks)\n", alloc);
        for (int pass = 0; pass < 2; ++pass)
            for (int V : {1, 2, 4}) {
                if (pass == 0) { for (int s : orders) (void)(V == 1 ? run<1>(in, out, nvec, s) : V == 2 ? run<2>(in, out, nvec, s) : run<4>(in, out, nvec, s)); continue; }
                printf("V=%d (%2d KiB per block):", V, 4 * V);
                for (int s : orders) {
                    const float ms = V == 1 ? run<1>(in, out, nvec, s) : V == 2 ? run<2>(in, out, nvec, s) : run<4>(in, out, nvec, s);
                    printf("  s=%2d %.4f", s, 2.0 * bytes / (ms * 1e-3) / 8e12);
                }
                printf("\n");
            }
    }
    return 0;
}
