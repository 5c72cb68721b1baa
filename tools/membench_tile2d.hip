// tools/membench_tile2d.hip -- round 4: what a TILE-shaped 2-D walk reaches with no arithmetic, against the strip walk of
// csrc/sg_2d_roll.hip (tools/membench_lockstep.hip: 0.63-0.71 of the HBM roofline bare) and a flat copy (0.79).
// Shape under test: a short-lived wave owns R output rows x 256 loaded columns.  It issues ALL its R + 2n row loads (16 B per lane each,
// rows one frame pitch apart) back to back into registers, then stores its R rows and exits -- the 1-D kernel's life cycle (8 KiB tiles,
// 0.77) turned by 90 degrees.  The 2n halo rows are read AGAIN by the tile below: HBM sees them once if the second read hits L2 / the
// Infinity Cache, which depends on which tiles run close together -- hence the tile orders below.
//   R      output rows per tile (16: 30 rows x 4 VGPRs in flight per lane, 1.875 x row reads; 32: 46 rows, 1.44 x)
//   SWC    stored columns per strip: 240 (one load per row, 18 strips per 4096 columns, the rolling kernel's layout at n = 7, 8) or
//          256 (a second, 4-lane load of the 16 columns beyond the KiB: 16 strips, stores on whole lines)
//   order  0 = strips fastest, then bands, then frames (launch order = address order of whole bands);  1 = bands fastest (a column of tiles)
//   xcd    0 = tiles in launch order (neighbouring tiles land on different XCDs);  1 = every XCD takes a contiguous eighth of the order
//   WPB    waves per block (no co-operation: only changes how waves are dealt to CUs);  cap = blocks per CU wanted (dynamic LDS pad), 0 = none
//   hipcc --offload-arch=gfx950 -O3 -o tools/membench_tile2d tools/membench_tile2d.hip && tools/membench_tile2d
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

__global__ __launch_bounds__(256) void k_flat(const v4f *__restrict__ in, v4f *__restrict__ out, size_t nvec)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nvec) __builtin_nontemporal_store(__builtin_bit_cast(u4, in[i]), (u4 *)(out + i));
}

struct Shape {
    int cols, rows;
    unsigned strips, bands, total;   // per frame; total = frames * strips * bands
    int order, xcd, nts;
};

constexpr int kHalo = 14;            // 2n at n = 7

// out[y] = in[y] + 1e-30 * (in[y - 7] + in[y + 7]): every loaded row is used, row r of the tile needs rows <= r + 14 (so the waits are counted)
template <int R, int SWC, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_tile(const float *__restrict__ in, float *__restrict__ out, const Shape s)
{
    extern __shared__ float occupancy_pad[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nblk = gridDim.x;
    const unsigned blk = s.xcd ? (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const unsigned t = blk * WPB + wv;
    if (t >= s.total) return;
    unsigned strip, band, img;
    if (s.order == 0) { strip = t % s.strips; const unsigned ib = t / s.strips; band = ib % s.bands; img = ib / s.bands; }
    else              { band = t % s.bands; const unsigned ib = t / s.bands; strip = ib % s.strips; img = ib / s.strips; }
    const int y0 = (int)band * R;
    int cx = (int)strip * SWC - 8;                                     // first loaded column
    if (cx < 0) cx = 0;
    const int span = SWC == 256 ? 272 : 256;
    if (cx + span > s.cols) cx = s.cols - span;
    const size_t frame = (size_t)s.cols * s.rows;
    const float *src = in + img * frame + cx + lane * 4;
    float *dst = out + img * frame + cx + lane * 4;
    const int c0 = cx + lane * 4;
    const bool keep = c0 >= (int)strip * SWC && c0 < (int)(strip + 1) * SWC && c0 < s.cols;
    v4f tile[R + kHalo];
    v4f tail[R + kHalo];
#pragma unroll
    for (int i = 0; i < R + kHalo; ++i) {
        int y = y0 - kHalo / 2 + i;
        y = y < 0 ? 0 : (y >= s.rows ? s.rows - 1 : y);
        tile[i] = *(const v4f *)(src + (size_t)y * s.cols);
        if constexpr (SWC == 256) { if (lane < 4) tail[i] = *(const v4f *)(src + (size_t)y * s.cols + 256); else tail[i] = v4f{0, 0, 0, 0}; }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        v4f v = tile[r + kHalo / 2] + 1e-30f * (tile[r] + tile[r + kHalo]);
        if constexpr (SWC == 256) v += 1e-30f * (tail[r] + tail[r + kHalo] + tail[r + kHalo / 2]);
        const int y = y0 + r;
        if (keep && y < s.rows) {
            if (s.nts) __builtin_nontemporal_store(__builtin_bit_cast(u4, v), (u4 *)(dst + (size_t)y * s.cols));
            else *(v4f *)(dst + (size_t)y * s.cols) = v;
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

static int kFrames = 64;
static const int kCols = 4096, kRows = 4096;
static float *g_in, *g_out;

template <int R, int SWC, int WPB>
static void run(int order, int xcd, int cap, int nts = 1)
{
    Shape s;
    s.cols = kCols; s.rows = kRows;
    s.strips = (unsigned)((kCols + SWC - 1) / SWC);
    s.bands = (unsigned)((kRows + R - 1) / R);
    s.total = (unsigned)kFrames * s.strips * s.bands;
    s.order = order; s.xcd = xcd; s.nts = nts;
    const void *fn = reinterpret_cast<const void *>(k_tile<R, SWC, WPB>);
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, fn));
    const size_t lds = cap > 0 ? (size_t)(160 * 1024 / cap) - 1024 : 0;
    if (lds > 65536) CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned grid = (s.total + WPB - 1) / WPB;
    grid = (grid + 7u) & ~7u;
    const double ms = time_ms([&] { hipLaunchKernelGGL((k_tile<R, SWC, WPB>), dim3(grid), dim3(64 * WPB), lds, 0, g_in, g_out, s); });
    const double bytes = 2.0 * kFrames * (double)kCols * kRows * 4;
    printf("tile R=%2d SWC=%3d WPB=%d order=%d xcd=%d cap=%d nts=%d vgpr=%3d grid=%7u : %7.3f ms  %6.1f GB/s  (%.3f of 8 TB/s)\n", R, SWC, WPB, order, xcd, cap, nts,
           fa.numRegs, grid, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    if (argc > 1) kFrames = atoi(argv[1]);
    const int only = argc > 2 ? atoi(argv[2]) : -1;                 // run one configuration (for rocprofv3 --pmc passes)
    CK(hipEventCreate(&ev_a)); CK(hipEventCreate(&ev_b));
    const size_t bytes = (size_t)kFrames * kCols * kRows * 4;
    CK(hipMalloc(&g_in, bytes)); CK(hipMalloc(&g_out, bytes));
    hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, (v4f *)g_in, bytes / 16);
    hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, (v4f *)g_out, bytes / 16);
    CK(hipDeviceSynchronize());
    if (only < 0) {
        const size_t nvec = bytes / 16;
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_flat, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, (const v4f *)g_in, (v4f *)g_out, nvec); });
        printf("flat copy, one vector per thread: %7.3f ms  %6.1f GB/s  (%.3f of 8 TB/s)\n", ms, 2.0 * bytes / ms / 1e6, 2.0 * bytes / ms / 1e6 / 8000.0);
    }
    if (only == 0) { run<16, 240, 4>(0, 0, 0); return 0; }
    if (only == 1) { run<16, 240, 4>(0, 1, 0); return 0; }
    if (only == 2) { run<32, 240, 4>(0, 1, 0); return 0; }
    if (only == 3) { run<16, 256, 4>(0, 1, 0); return 0; }
    // tile height x order x XCD grouping, 4 waves per block, occupancy as the registers allow
    for (int order = 0; order < 2; ++order)
        for (int xcd = 0; xcd < 2; ++xcd) {
            run<8, 240, 4>(order, xcd, 0);
            run<16, 240, 4>(order, xcd, 0);
            run<24, 240, 4>(order, xcd, 0);
            run<32, 240, 4>(order, xcd, 0);
        }
    // one wave per block; plain stores; whole-line strips
    run<16, 240, 1>(0, 0, 0);
    run<16, 240, 1>(0, 1, 0);
    run<32, 240, 1>(0, 1, 0);
    run<16, 240, 4>(0, 0, 0, 0);
    run<16, 240, 4>(0, 1, 0, 0);
    run<16, 256, 4>(0, 0, 0);
    run<16, 256, 4>(0, 1, 0);
    run<32, 256, 4>(0, 1, 0);
    // fewer resident blocks (what the real kernel's registers will allow: 3 or 2 waves per SIMD)
    for (int cap = 3; cap >= 1; --cap) {
        run<16, 240, 4>(0, 0, cap);
        run<16, 240, 4>(0, 1, cap);
        run<32, 240, 4>(0, 1, cap);
    }
    return 0;
}
