// tools/membench2d.hip -- ceiling of the 2-D rolling kernel's ACCESS SHAPE on one MI355X, without its arithmetic.
// The rolling kernel (csrc/sg_2d_roll.hip) gives a wave a 256-column strip of a frame and walks it row by row: one 1 KiB load and
// one ~1 KiB store per row, rows one frame stride (16 KiB at 4096 columns) apart.  At half windows <= 5 it sits at 0.52-0.55 of
// the HBM roofline with its waves parked on s_waitcnt 78 % of the time, and neither deeper prefetch nor wider blocks moved it.
// This program copies frames with exactly that walk -- and with the knobs the kernel could turn -- to see what the shape allows:
//   P   rows in flight per wave (1 = the kernel at n <= 5)            W   16-byte vectors per lane per row (strip = 256*W columns)
//   SWC stored columns per strip (256*W = disjoint strips; 248 = the kernel's overlapping strips at n = 2: reads straddle lines)
//   persistent grid vs one item per wave, band height, waves per block
//   hipcc --offload-arch=gfx950 -O3 -o tools/membench2d tools/membench2d.hip && tools/membench2d
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_init(v4f *p, size_t nvec)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull + 0x5A17601Aull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = v4f{(float)(z & 0xffff) * 1e-4f, (float)((z >> 16) & 0xffff) * 1e-4f, (float)((z >> 32) & 0xffff) * 1e-4f, (float)(z >> 48) * 1e-4f};
    }
}

struct Shape {
    int cols, rows;          // frame
    unsigned strips, bands;  // per frame
    int band_rows, swc;      // rows per band, stored columns per strip
    int lead;                // loaded columns ahead of the first stored one (-1: centre the stored columns in the loaded ones)
    unsigned total;          // items = frames * strips * bands
};

// WPB waves per block; a wave walks one item (strip x band) per round.  Loads go through a register ring of P rows; the row
// index is clamped instead of tested so the loop has no branch besides its own back edge (waitcnt counts stay exact).
template <int P, int W, int WPB, int NTS>
__global__ __launch_bounds__(64 * WPB) void k_walk(const float *__restrict__ in, float *__restrict__ out, const Shape s)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nblk = gridDim.x;
    const unsigned blk = (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3);
    const unsigned nwaves = nblk * WPB;
    const size_t frame = (size_t)s.cols * s.rows;
    for (unsigned item = blk * WPB + wv; item < s.total; item += nwaves) {
        const unsigned strip = item % s.strips, ib = item / s.strips;
        const unsigned band = ib % s.bands, img = ib / s.bands;
        const int y0 = (int)band * s.band_rows;
        const int nrow = s.rows - y0 < s.band_rows ? s.rows - y0 : s.band_rows;
        const int lead = s.lead < 0 ? (256 * W - s.swc) / 2 : s.lead;
        int cx = (int)strip * s.swc - lead;                            // first loaded column (overlapping strips start early)
        if (cx < 0) cx = 0;
        if (cx + 256 * W > s.cols) cx = s.cols - 256 * W;
        const float *src = in + img * frame + (size_t)y0 * s.cols + cx + lane * 4;
        float *dst = out + img * frame + (size_t)y0 * s.cols + cx + lane * 4;
        const int c0 = cx + lane * 4;                                  // W == 1 when strips overlap
        const bool keep = W * 256 == s.swc || (c0 >= (int)strip * s.swc && c0 < (int)(strip + 1) * s.swc && c0 < s.cols);
        v4f ring[P][W];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int r = p < nrow ? p : nrow - 1;
#pragma unroll
            for (int w = 0; w < W; ++w) ring[p][w] = *(const v4f *)(src + (size_t)r * s.cols + 256 * w);
        }
        for (int y = 0; y < nrow; y += P) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                v4f cur[W];
#pragma unroll
                for (int w = 0; w < W; ++w) cur[w] = ring[p][w];
                int r = y + p + P; r = r < nrow ? r : nrow - 1;
#pragma unroll
                for (int w = 0; w < W; ++w) ring[p][w] = *(const v4f *)(src + (size_t)r * s.cols + 256 * w);
                if (y + p < nrow && keep) {
#pragma unroll
                    for (int w = 0; w < W; ++w) {
                        if constexpr (NTS) __builtin_nontemporal_store(cur[w], (v4f *)(dst + (size_t)(y + p) * s.cols + 256 * w));
                        else *(v4f *)(dst + (size_t)(y + p) * s.cols + 256 * w) = cur[w];
                    }
                }
            }
        }
    }
}

static hipEvent_t ev_a, ev_b;
template <typename F>
static double time_ms(F launch, int iters = 7)
{
    launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(ev_a));
        launch();
        CK(hipEventRecord(ev_b));
        CK(hipEventSynchronize(ev_b));
        float t; CK(hipEventElapsedTime(&t, ev_a, ev_b));
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

static const int kFrames = 64, kCols = 4096, kRows = 4096;
static float *g_in, *g_out;

template <int P, int W, int WPB, int NTS>
static void run(int swc, unsigned bands, bool persistent, int blocks_per_cu, int lead = -1)
{
    Shape s;
    s.lead = lead;
    s.cols = kCols; s.rows = kRows; s.swc = swc;
    s.strips = (unsigned)((kCols + swc - 1) / swc);
    s.band_rows = (kRows + (int)bands - 1) / (int)bands;
    s.bands = (unsigned)((kRows + s.band_rows - 1) / s.band_rows);
    s.total = (unsigned)kFrames * s.strips * s.bands;
    unsigned grid = persistent ? 256u * (unsigned)blocks_per_cu : (s.total + WPB - 1) / WPB;
    if ((unsigned long long)grid * WPB > s.total) grid = (s.total + WPB - 1) / WPB;
    grid = (grid + 7u) & ~7u;
    const double ms = time_ms([&] { hipLaunchKernelGGL((k_walk<P, W, WPB, NTS>), dim3(grid), dim3(64 * WPB), 0, 0, g_in, g_out, s); });
    const double bytes = 2.0 * kFrames * (double)kCols * kRows * 4;
    printf("P=%d W=%d WPB=%2d nt_st=%d swc=%4d lead=%2d bands=%3u (%4d rows) %-10s grid=%6u : %7.3f ms  %6.1f GB/s  (%.3f of 8 TB/s)\n", P, W, WPB, NTS, swc, lead < 0 ? (256 * W - swc) / 2 : lead,
           s.bands, s.band_rows, persistent ? "persistent" : "item/wave", grid, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
    fflush(stdout);
}

int main()
{
    CK(hipEventCreate(&ev_a)); CK(hipEventCreate(&ev_b));
    const size_t bytes = (size_t)kFrames * kCols * kRows * 4;
    CK(hipMalloc(&g_in, bytes)); CK(hipMalloc(&g_out, bytes));
    hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, (v4f *)g_in, bytes / 16);
    hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, (v4f *)g_out, bytes / 16);
    CK(hipDeviceSynchronize());
    printf("copy of %d frames %dx%d fp32 with the rolling kernel's walk; GB/s counts bytes read + written\n", kFrames, kCols, kRows);
    // 1. the kernel's shape at n <= 5: one row in flight, 4 waves per block, persistent, 4 bands
    run<1, 1, 4, 1>(256, 4, true, 4);
    run<1, 1, 4, 1>(248, 4, true, 4);
    run<1, 1, 4, 0>(256, 4, true, 4);
    // 1b. where overlapping strips lose: line-straddling loads, partial-line stores, or the streaming hint on those stores.
    //     One item per wave and 64 bands, so that the strip count (17 or 19 per row instead of 16) does not leave a ragged last round
    run<1, 1, 4, 1>(256, 64, false, 4, 0);       // disjoint, everything on 128-byte lines
    run<1, 1, 4, 1>(248, 64, false, 4, 4);       // the kernel at n <= 4: 992-byte stores, loads 16 bytes early
    run<1, 1, 4, 0>(248, 64, false, 4, 4);       //   ... with plain stores (partial lines may merge in L2)
    run<1, 1, 4, 1>(240, 64, false, 4, 8);       // the kernel at n = 5..8: 960-byte stores on 64-byte boundaries
    run<1, 1, 4, 0>(240, 64, false, 4, 8);
    run<1, 1, 4, 1>(224, 64, false, 4, 16);      // 896-byte stores on 128-byte lines, loads 64 bytes early
    run<1, 1, 4, 1>(224, 64, false, 4, 0);       // stores and loads on 128-byte lines (stored columns at the left of the loaded ones)
    run<1, 1, 4, 1>(192, 64, false, 4, 32);      // all on lines, a third more loads
    run<4, 1, 4, 1>(248, 64, false, 4, 4);
    run<4, 1, 4, 1>(224, 64, false, 4, 16);
    // 2. rows in flight
    run<2, 1, 4, 1>(256, 4, true, 4);
    run<4, 1, 4, 1>(256, 4, true, 4);
    run<8, 1, 4, 1>(256, 4, true, 4);
    run<4, 1, 4, 1>(248, 4, true, 4);
    // 3. wider strips
    run<1, 2, 4, 1>(512, 8, true, 4);
    run<2, 2, 4, 1>(512, 8, true, 4);
    run<4, 2, 4, 1>(512, 8, true, 4);
    run<1, 4, 4, 1>(1024, 16, true, 4);
    run<2, 4, 4, 1>(1024, 16, true, 4);
    // 4. waves per block (a block's row step = one contiguous run)
    run<1, 1, 16, 1>(256, 4, true, 1);
    run<4, 1, 16, 1>(256, 4, true, 1);
    // 5. one item per wave, short bands (block order instead of a persistent grid)
    run<1, 1, 4, 1>(256, 32, false, 4);
    run<1, 1, 4, 1>(256, 64, false, 4);
    run<1, 1, 4, 1>(256, 128, false, 4);
    run<4, 1, 4, 1>(256, 32, false, 4);
    run<4, 1, 4, 1>(256, 64, false, 4);
    run<4, 1, 4, 1>(256, 128, false, 4);
    run<4, 1, 4, 1>(256, 256, false, 4);
    run<8, 1, 4, 1>(256, 128, false, 4);
    run<4, 2, 4, 1>(512, 128, false, 4);
    run<4, 1, 16, 1>(256, 128, false, 1);
    // 6. fewer resident waves with deep rings
    run<8, 1, 4, 1>(256, 4, true, 2);
    run<8, 1, 4, 1>(256, 8, true, 1);
    return 0;
}
