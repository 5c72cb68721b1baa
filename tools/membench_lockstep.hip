// tools/membench_lockstep.hip -- round 3: what limits the strip walk of the 2-D rolling kernel (and of the stream block push) to
// 0.63-0.67 of the HBM roofline when a flat copy reaches 0.81?  Hypothesis tested here: every wave of a strip walk is its own
// sequential stream through memory (3072 waves x read + write), while short-lived waves dispatched in address order look like ONE
// stream to the DRAM.  So: make the waves of a block cover a whole 16 KiB frame row and keep them in step with a barrier every
// BAR rows (one contiguous 16 KiB read and write per block and row step), vary how many such blocks are resident, and compare with
// the free-running strip walk and the flat copy in the same process.
//   hipcc --offload-arch=gfx950 -O3 -o tools/membench_lockstep tools/membench_lockstep.hip && tools/membench_lockstep
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_init(v4f *p, size_t nvec)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull + 0x5A17601Aull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z ^= z >> 31;
        p[i] = v4f{(float)(z & 0xffff) * 1e-4f, (float)((z >> 16) & 0xffff) * 1e-4f, (float)((z >> 32) & 0xffff) * 1e-4f, (float)(z >> 48) * 1e-4f};
    }
}

// flat copy, one 16-byte vector per thread (the guide's shape)
__global__ __launch_bounds__(256) void k_flat(const v4f *__restrict__ in, v4f *__restrict__ out, size_t nvec)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nvec) __builtin_nontemporal_store(__builtin_bit_cast(u4, in[i]), (u4 *)(out + i));
}

__device__ int g_pitch_dev;                                          // 0 = the frame width; set by main() for the pitch experiment
// WPB waves per block, side by side: the block covers 256*WPB columns.  An item = (frame, column group, band of rows).  P rows in
// flight per wave.  BAR > 0: __syncthreads() every BAR rows.  XCD: 1 = blocks that share an XCD take neighbouring items.
template <int P, int WPB, int BAR>
__global__ __launch_bounds__(64 * WPB) void k_rows(const float *__restrict__ in, float *__restrict__ out, int cols_, int rows, int band_rows,
                                                    unsigned groups, unsigned bands, unsigned total, int xcd)
{
    const int cols = g_pitch_dev ? g_pitch_dev : cols_;              // row pitch in floats (frames are rows x pitch)
    extern __shared__ float occupancy_pad[];                        // dynamic LDS only limits how many blocks a CU holds
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned nblk = gridDim.x;
    const unsigned blk = xcd ? (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    for (unsigned item = blk; item < total; item += nblk) {
        const unsigned g = item % groups, ib = item / groups, band = ib % bands, img = ib / bands;
        const int y0 = (int)band * band_rows;
        const int nrow = rows - y0 < band_rows ? rows - y0 : band_rows;
        const size_t base = (size_t)img * cols * rows + (size_t)y0 * cols + (size_t)g * 256 * WPB + wv * 256 + lane * 4;
        const float *src = in + base;
        float *dst = out + base;
        v4f ring[P];
#pragma unroll
        for (int p = 0; p < P; ++p) ring[p] = *(const v4f *)(src + (size_t)(p < nrow ? p : nrow - 1) * cols);
        for (int y = 0; y < nrow; y += P) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const v4f cur = ring[p];
                int r = y + p + P; r = r < nrow ? r : nrow - 1;
                ring[p] = *(const v4f *)(src + (size_t)r * cols);
                if (y + p < nrow) __builtin_nontemporal_store(__builtin_bit_cast(u4, cur), (u4 *)(dst + (size_t)(y + p) * cols));
                if constexpr (BAR > 0) { if ((y + p) % BAR == BAR - 1) __syncthreads(); }
            }
        }
    }
    (void)occupancy_pad;
}

static hipEvent_t ev_a, ev_b;
template <typename F>
static double time_ms(F launch, int iters = 7)
{
    launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(ev_a)); launch(); CK(hipEventRecord(ev_b)); CK(hipEventSynchronize(ev_b));
        float t; CK(hipEventElapsedTime(&t, ev_a, ev_b)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

static const int kFrames = 64, kCols = 4096, kRows = 4096;
static float *g_in, *g_out;

// blocks_per_cu: resident blocks per CU wanted (enforced through dynamic LDS: 160 KB / blocks_per_cu); persistent grid when > 0,
// one item per block (grid = items) when 0
template <int P, int WPB, int BAR>
static void run(unsigned bands, int blocks_per_cu, int xcd)
{
    const unsigned groups = (unsigned)(kCols / (256 * WPB));
    const int band_rows = (kRows + (int)bands - 1) / (int)bands;
    const unsigned total = (unsigned)kFrames * groups * bands;
    const size_t lds = blocks_per_cu > 0 ? (size_t)(160 * 1024 / blocks_per_cu) - 1024 : 0;
    if (lds > 65536) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_rows<P, WPB, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned grid = blocks_per_cu > 0 ? 256u * (unsigned)blocks_per_cu : total;
    if (grid > total) grid = total;
    grid = (grid + 7u) & ~7u;
    const double ms = time_ms([&] { hipLaunchKernelGGL((k_rows<P, WPB, BAR>), dim3(grid), dim3(64 * WPB), lds, 0, g_in, g_out, kCols, kRows, band_rows, groups, bands, total, xcd); });
    const double bytes = 2.0 * kFrames * (double)kCols * kRows * 4;
    printf("P=%d WPB=%2d BAR=%2d bands=%4u (%4d rows) %s blocks/CU=%d xcd=%d grid=%6u : %7.3f ms  %6.1f GB/s  (%.3f of 8 TB/s)\n", P, WPB, BAR, bands, band_rows,
           blocks_per_cu > 0 ? "persistent" : "item/block", blocks_per_cu, xcd, grid, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    CK(hipEventCreate(&ev_a)); CK(hipEventCreate(&ev_b));
    if (argc > 1 && !strcmp(argv[1], "pitch")) {
        // Is it the power-of-two row pitch?  A wave's rows in flight are 16 KiB apart: the same L1 / L2 sets.  Same walk, same 4096 columns
        // copied, frames laid out with a longer pitch.
        const int pitches[] = {4096, 4128, 4160, 4224, 4352, 4608, 5120};
        const size_t big = (size_t)kFrames * 5120 * kRows * 4;
        CK(hipMalloc(&g_in, big)); CK(hipMalloc(&g_out, big));
        hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, (v4f *)g_in, big / 16);
        hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, (v4f *)g_out, big / 16);
        CK(hipDeviceSynchronize());
        for (int p : pitches) {
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_pitch_dev), &p, sizeof(int)));
            printf("-- row pitch %d floats (%d bytes)\n", p, p * 4);
            run<4, 4, 0>(16, 3, 0);
            run<4, 4, 0>(16, 0, 0);
            run<8, 4, 0>(16, 1, 0);
            run<8, 16, 8>(16, 1, 0);
        }
        return 0;
    }
    const size_t bytes = (size_t)kFrames * kCols * kRows * 4;
    CK(hipMalloc(&g_in, bytes)); CK(hipMalloc(&g_out, bytes));
    hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, (v4f *)g_in, bytes / 16);
    hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, (v4f *)g_out, bytes / 16);
    CK(hipDeviceSynchronize());
    {
        const size_t nvec = bytes / 16;
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_flat, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, (const v4f *)g_in, (v4f *)g_out, nvec); });
        printf("flat copy, one vector per thread: %7.3f ms  %6.1f GB/s  (%.3f of 8 TB/s)\n", ms, 2.0 * bytes / ms / 1e6, 2.0 * bytes / ms / 1e6 / 8000.0);
    }
    // the kernel's walk: 4 waves per block, free running, 3 blocks per CU, 16 bands
    run<4, 4, 0>(16, 3, 1);
    run<4, 4, 0>(16, 3, 0);
    run<4, 4, 0>(16, 0, 1);
    run<4, 4, 0>(16, 0, 0);
    run<4, 4, 0>(64, 0, 0);
    run<4, 4, 0>(256, 0, 0);
    // whole-row blocks (16 waves = 4096 columns = 16 KiB per row step), free running and in step
    run<4, 16, 0>(16, 1, 1);
    run<4, 16, 0>(16, 1, 0);
    run<4, 16, 1>(16, 1, 1);
    run<4, 16, 1>(16, 1, 0);
    run<4, 16, 4>(16, 1, 0);
    run<8, 16, 4>(16, 1, 0);
    run<8, 16, 8>(16, 1, 0);
    run<4, 16, 1>(16, 2, 0);
    run<4, 16, 4>(16, 2, 0);
    run<8, 16, 8>(16, 2, 0);
    run<4, 16, 4>(64, 0, 0);
    run<4, 16, 4>(256, 0, 0);
    run<8, 16, 8>(128, 0, 0);
    run<8, 16, 0>(128, 0, 0);
    // fewer resident waves (is it the number of concurrent streams?)
    run<8, 4, 0>(16, 1, 0);
    run<8, 4, 0>(16, 2, 0);
    run<8, 4, 0>(16, 4, 0);
    run<8, 8, 0>(16, 1, 0);
    run<8, 8, 8>(16, 1, 0);
    run<8, 8, 8>(16, 2, 0);
    return 0;
}
