// tools/stamp_1d.hip -- diagnostic build of the 1-D kernel with s_memtime stamps per phase.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DSG_STAMPS -Iinclude -Isavitzky-golay-filter_amd/csrc \
//         -o tools/stamp_1d tools/stamp_1d.hip
// Never timed as a product number: the stamps serialise the phases.  Read the SHARES.
#include "sg_k1d.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename T, int N>
static void run(const T *in, T *out, unsigned channels, unsigned length, unsigned grid)
{
    sg::Job1D job;
    memset(&job, 0, sizeof(job));
    job.in = in; job.out = out; job.in_ld = length; job.out_ld = length; job.length = length;
    const unsigned TW = 64 * sg::vectors_per_lane(sizeof(T), N) * (16 / sizeof(T));
    const unsigned tpc_ = (length + TW - 1) / TW;
    sg::set_tiles_per_channel(job, tpc_);
    job.total_tiles = channels * job.tiles_per_channel;
    job.store_lo = 0; job.store_hi = length; job.out_shift = 0; job.dt_inv = 1.0f;
    job.flags = 1u | sg::JOB_VEC_IN | sg::JOB_VEC_OUT;
    sg::Taps taps;
    memset(&taps, 0, sizeof(taps));
    for (int k = 0; k < 2 * N + 1; ++k) taps.w[k] = 1.0f / (2 * N + 1);
    grid = ((job.total_tiles + 3) / 4 + 7) & ~7u;             // round 2: one tile per wave, blocks in order (the argument is ignored)
    unsigned long long *d_st;
    CK(hipMalloc(&d_st, 64 * 8 * 8));
    CK(hipMemset(d_st, 0, 64 * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(sg::g_stamps), &d_st, sizeof(d_st)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((sg::sg1d_center_kernel<T, N>), dim3(grid), dim3(256), 0, 0, job, taps);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    }
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> st(64 * 8);
    CK(hipMemcpy(st.data(), d_st, 64 * 8 * 8, hipMemcpyDeviceToHost));
    printf("%s N=%d grid=%u: %.3f ms (stamped build)\n", sizeof(T) == 4 ? "f32" : "f64", N, grid, ms);
    printf("  one tile of block 8 / wave 1:  load+stage   (sync)   compute   store   | total (cycles)\n");
    for (int it = 0; it < 12; ++it) {
        const unsigned long long *s = &st[it * 8];
        if (!s[4]) break;
        printf("  %3d: %10llu %10llu %10llu %10llu | %10llu\n", it, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[4] - s[0]);
    }
}

int main()
{
    const unsigned channels = 4096, length = 1u << 20;
    float *in, *out;
    CK(hipMalloc(&in, (size_t)channels * length * 4)); CK(hipMalloc(&out, (size_t)channels * length * 4));
    CK(hipMemset(in, 0, (size_t)channels * length * 4));
    run<float, 5>(in, out, channels, length, 1024);
    run<float, 32>(in, out, channels, length, 1024);
    run<double, 32>((const double *)in, (double *)out, channels / 2, length, 1024);
    run<double, 16>((const double *)in, (double *)out, channels / 2, length, 1024);
    run<double, 5>((const double *)in, (double *)out, channels / 2, length, 1024);
    return 0;
}
