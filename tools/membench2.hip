// tools/membench2.hip -- what a read+write stream can reach on one MI355X, and with which access shape.
// Round-2 question: tools/membench.hip measured 5.46 TB/s for a copy while
// MI355X_MICROARCH.md lists 6.29 TB/s for "a float4 copy".  This program times every copy shape
// named there -- hipMemcpyDtoDAsync, one vector per thread (non-persistent), U vectors per thread,
// persistent tiles, with and without nontemporal hints, at 1/2/4/8 waves per SIMD, at three buffer
// sizes, with the output buffer displaced by a few deltas -- plus read-only and write-only streams.
//   hipcc --offload-arch=gfx950 -O3 -o tools/membench2 tools/membench2.hip && tools/membench2 [--quick]
// "total GB/s" always counts bytes read + bytes written (the same convention as roofline.achieved).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

template <int NT> __device__ __forceinline__ v4f ld(const v4f *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}
template <int NT> __device__ __forceinline__ void st(v4f *p, v4f v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

extern __shared__ char dyn_lds[];   // only there to limit blocks per CU

// noisy fill (data-dependent power: all-zero buffers run a few % faster)
__global__ __launch_bounds__(256) void k_init(v4f *p, size_t nvec)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull + 0x5A17601Aull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        v4f v = {(float)(z & 0xffff) * 1e-4f, (float)((z >> 16) & 0xffff) * 1e-4f, (float)((z >> 32) & 0xffff) * 1e-4f, (float)(z >> 48) * 1e-4f};
        p[i] = v;
    }
}

// MODE 0 copy, 1 read only, 2 write only.  U vectors per thread, block covers 256*U contiguous vectors,
// a wave-instruction touches 1 KiB contiguous.  Non-persistent: grid = nvec / (256*U).
template <int U, int NTL, int NTS, int MODE>
__global__ __launch_bounds__(256) void k_block(const v4f *__restrict__ in, v4f *__restrict__ out, size_t nvec)
{
    const size_t b0 = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    v4f v[U];
    v4f acc = {0, 0, 0, 0};
    if constexpr (MODE != 2) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NTL>(in + b0 + 256 * u);
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = v4f{(float)threadIdx.x, 1.f, 2.f, (float)u};
    }
    if constexpr (MODE != 1) {
#pragma unroll
        for (int u = 0; u < U; ++u) st<NTS>(out + b0 + 256 * u, v[u]);
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
        if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc;
    }
    (void)nvec;
}

// persistent waves, one tile of U KiB per wave per round, next tile prefetched into registers before this
// one is stored (the 1-D kernel's shape); XCD-aware block remap as in sg1d_center_kernel
template <int U, int NTL, int NTS>
__global__ __launch_bounds__(256) void k_persist(const v4f *__restrict__ in, v4f *__restrict__ out, unsigned ntiles)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nblk = gridDim.x;
    const unsigned blk = (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3);
    const unsigned nwaves = nblk * 4;
    v4f v[U], nx[U];
    unsigned tile = blk * 4 + wave;
    if (tile < ntiles) {
        const v4f *src = in + (size_t)tile * (64 * U);
#pragma unroll
        for (int s = 0; s < U; ++s) nx[s] = ld<NTL>(src + lane + 64 * s);
    }
    for (; tile < ntiles; tile += nwaves) {
#pragma unroll
        for (int s = 0; s < U; ++s) v[s] = nx[s];
        if (tile + nwaves < ntiles) {
            const v4f *s2 = in + (size_t)(tile + nwaves) * (64 * U);
#pragma unroll
            for (int s = 0; s < U; ++s) nx[s] = ld<NTL>(s2 + lane + 64 * s);
        }
        v4f *dst = out + (size_t)tile * (64 * U);
#pragma unroll
        for (int s = 0; s < U; ++s) st<NTS>(dst + lane + 64 * s, v[s]);
    }
}

// half of the blocks only read, the other half only write (different halves of the buffers): is it the
// read/write mix at the memory that costs, or the mix inside a wave?
template <int U>
__global__ __launch_bounds__(256) void k_split(const v4f *__restrict__ in, v4f *__restrict__ out, size_t nvec_half)
{
    const bool writer = blockIdx.x & 1;
    const size_t b0 = (size_t)(blockIdx.x >> 1) * (256 * U) + threadIdx.x;
    if (writer) {
#pragma unroll
        for (int u = 0; u < U; ++u) st<1>(out + nvec_half + b0 + 256 * u, v4f{(float)threadIdx.x, 1.f, 2.f, (float)u});
    } else {
        v4f acc = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) acc += ld<1>(in + b0 + 256 * u);
        if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc;
    }
}

// one vector per thread with BS threads per block (the block -> XCD round robin then deals the buffer out in BS*16-byte
// pieces), optionally with the XCD remap of the 1-D kernel (8 fronts instead of one)
template <int BS, int REMAP>
__global__ __launch_bounds__(BS) void k_one(const v4f *__restrict__ in, v4f *__restrict__ out)
{
    const unsigned nblk = gridDim.x;
    const unsigned blk = REMAP ? (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const size_t i = (size_t)blk * BS + threadIdx.x;
    st<1>(out + i, ld<1>(in + i));
}

// a wave owns U contiguous KiB (the tile shape of the 1-D kernel), 4 waves per block, one tile per wave, non-persistent
template <int U, int REMAP>
__global__ __launch_bounds__(256) void k_wavetile(const v4f *__restrict__ in, v4f *__restrict__ out)
{
    const unsigned nblk = gridDim.x;
    const unsigned blk = REMAP ? (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const size_t b0 = ((size_t)blk * 4 + (threadIdx.x >> 6)) * (64 * U) + (threadIdx.x & 63);
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<1>(in + b0 + 64 * u);
#pragma unroll
    for (int u = 0; u < U; ++u) st<1>(out + b0 + 64 * u, v[u]);
}

static hipEvent_t ev_a, ev_b;
template <typename F>
static double time_ms(F launch, int iters = 5)
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

static void report(const char *name, double ms, double bytes_moved)
{
    printf("%-64s %8.3f ms  total %7.1f GB/s  (%.1f %% of 8 TB/s)\n", name, ms, bytes_moved / ms / 1e6, bytes_moved / ms / 1e6 / 80.0);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const bool quick = argc > 1 && !strcmp(argv[1], "--quick");
    CK(hipEventCreate(&ev_a)); CK(hipEventCreate(&ev_b));
    const size_t GiB = (size_t)1 << 30;
    const size_t bytes = 16 * GiB;
    const size_t slack = 256u << 20;
    char *pool;
    CK(hipMalloc(&pool, 2 * bytes + 2 * slack));
    v4f *in = (v4f *)pool;
    v4f *out = (v4f *)(pool + bytes + slack);
    hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, in, (2 * bytes + 2 * slack) / 16);
    CK(hipDeviceSynchronize());
    const size_t nvec = bytes / 16;
    char name[256];

    // ---- 1. the named shapes at 16 GiB in + 16 GiB out ----
    report("hipMemcpyDtoDAsync 16 GiB", time_ms([&] { CK(hipMemcpyDtoDAsync((hipDeviceptr_t)out, (hipDeviceptr_t)in, bytes, 0)); }), 2.0 * bytes);
#define BLOCK(U, NTL, NTS, MODE, label, moved) \
    report(label, time_ms([&] { hipLaunchKernelGGL((k_block<U, NTL, NTS, MODE>), dim3((unsigned)(nvec / (256 * U))), dim3(256), 0, 0, in, out, nvec); }), moved)
    BLOCK(1, 0, 0, 0, "copy  1 vec/thread  non-persistent  plain", 2.0 * bytes);
    BLOCK(1, 1, 1, 0, "copy  1 vec/thread  non-persistent  nt ld + nt st", 2.0 * bytes);
    BLOCK(2, 1, 1, 0, "copy  2 vec/thread  non-persistent  nt ld + nt st", 2.0 * bytes);
    BLOCK(4, 0, 0, 0, "copy  4 vec/thread  non-persistent  plain", 2.0 * bytes);
    BLOCK(4, 1, 0, 0, "copy  4 vec/thread  non-persistent  nt ld", 2.0 * bytes);
    BLOCK(4, 0, 1, 0, "copy  4 vec/thread  non-persistent  nt st", 2.0 * bytes);
    BLOCK(4, 1, 1, 0, "copy  4 vec/thread  non-persistent  nt ld + nt st", 2.0 * bytes);
    BLOCK(8, 1, 1, 0, "copy  8 vec/thread  non-persistent  nt ld + nt st", 2.0 * bytes);
    BLOCK(16, 1, 1, 0, "copy 16 vec/thread  non-persistent  nt ld + nt st", 2.0 * bytes);
    BLOCK(4, 0, 0, 1, "read  4 vec/thread  non-persistent  plain", 1.0 * bytes);
    BLOCK(4, 1, 0, 1, "read  4 vec/thread  non-persistent  nt", 1.0 * bytes);
    BLOCK(4, 0, 0, 2, "write 4 vec/thread  non-persistent  plain", 1.0 * bytes);
    BLOCK(4, 0, 1, 2, "write 4 vec/thread  non-persistent  nt", 1.0 * bytes);
    BLOCK(8, 0, 1, 2, "write 8 vec/thread  non-persistent  nt", 1.0 * bytes);
    report("split: odd blocks write 8 GiB, even blocks read 8 GiB (nt)",
           time_ms([&] { hipLaunchKernelGGL((k_split<4>), dim3((unsigned)(nvec / 2 / (256 * 4)) * 2), dim3(256), 0, 0, in, out, nvec / 2); }), 1.0 * bytes);
    if (quick) return 0;

    // ---- 2. persistent tiles (the 1-D kernel's shape) ----
    for (int g : {1024, 2048, 4096}) {
        snprintf(name, sizeof name, "copy persistent 8 KiB tiles, prefetch, nt, grid %d", g);
        report(name, time_ms([&] { hipLaunchKernelGGL((k_persist<8, 1, 1>), dim3(g), dim3(256), 0, 0, in, out, (unsigned)(bytes / 8192)); }), 2.0 * bytes);
        snprintf(name, sizeof name, "copy persistent 8 KiB tiles, prefetch, plain, grid %d", g);
        report(name, time_ms([&] { hipLaunchKernelGGL((k_persist<8, 0, 0>), dim3(g), dim3(256), 0, 0, in, out, (unsigned)(bytes / 8192)); }), 2.0 * bytes);
        snprintf(name, sizeof name, "copy persistent 4 KiB tiles, prefetch, nt, grid %d", g);
        report(name, time_ms([&] { hipLaunchKernelGGL((k_persist<4, 1, 1>), dim3(g), dim3(256), 0, 0, in, out, (unsigned)(bytes / 4096)); }), 2.0 * bytes);
    }

    // ---- 3. waves per SIMD (blocks per CU limited through dynamic LDS): 8, 4, 2, 1 ----
    for (int bpc : {8, 4, 2, 1}) {
        const unsigned lds = bpc == 8 ? 0 : (unsigned)(160 * 1024 / bpc);
        CK(hipFuncSetAttribute((const void *)k_block<4, 1, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipFuncSetAttribute((const void *)k_block<16, 1, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipFuncSetAttribute((const void *)k_persist<8, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        snprintf(name, sizeof name, "copy  4 vec/thread nt, %d waves/SIMD", bpc);
        report(name, time_ms([&] { hipLaunchKernelGGL((k_block<4, 1, 1, 0>), dim3((unsigned)(nvec / 1024)), dim3(256), lds, 0, in, out, nvec); }), 2.0 * bytes);
        snprintf(name, sizeof name, "copy 16 vec/thread nt, %d waves/SIMD", bpc);
        report(name, time_ms([&] { hipLaunchKernelGGL((k_block<16, 1, 1, 0>), dim3((unsigned)(nvec / 4096)), dim3(256), lds, 0, in, out, nvec); }), 2.0 * bytes);
        snprintf(name, sizeof name, "copy persistent 8 KiB nt, %d waves/SIMD (grid %d)", bpc, 256 * bpc);
        report(name, time_ms([&] { hipLaunchKernelGGL((k_persist<8, 1, 1>), dim3(256 * bpc), dim3(256), lds, 0, in, out, (unsigned)(bytes / 8192)); }), 2.0 * bytes);
    }

    // ---- 4. buffer size (TLB reach / Infinity Cache): same kernel on 256 MiB, 1, 4, 16 GiB ----
    for (size_t sz : {GiB / 4, GiB, 4 * GiB, 16 * GiB}) {
        const size_t nv = sz / 16;
        snprintf(name, sizeof name, "copy  4 vec/thread nt, %5.2f GiB in + same out", (double)sz / GiB);
        report(name, time_ms([&] { hipLaunchKernelGGL((k_block<4, 1, 1, 0>), dim3((unsigned)(nv / 1024)), dim3(256), 0, 0, in, out, nv); }, 9), 2.0 * sz);
        snprintf(name, sizeof name, "read  4 vec/thread nt, %5.2f GiB", (double)sz / GiB);
        report(name, time_ms([&] { hipLaunchKernelGGL((k_block<4, 1, 0, 1>), dim3((unsigned)(nv / 1024)), dim3(256), 0, 0, in, out, nv); }, 9), 1.0 * sz);
    }

    // ---- 5. displacement of the output buffer relative to the input buffer ----
    for (size_t delta : {(size_t)0, (size_t)256, (size_t)4096, (size_t)65536, (size_t)1 << 20, (size_t)3 << 20, (size_t)(33u << 20) + 8192}) {
        v4f *o2 = (v4f *)(pool + bytes + slack / 2 + delta);
        snprintf(name, sizeof name, "copy  4 vec/thread nt, out displaced by %zu B", delta);
        report(name, time_ms([&] { hipLaunchKernelGGL((k_block<4, 1, 1, 0>), dim3((unsigned)(nvec / 1024)), dim3(256), 0, 0, in, o2, nvec); }), 2.0 * bytes);
    }
    // in place: every line is read and then written back (same DRAM page)
    report("copy  4 vec/thread nt, IN PLACE (out == in)", time_ms([&] { hipLaunchKernelGGL((k_block<4, 1, 1, 0>), dim3((unsigned)(nvec / 1024)), dim3(256), 0, 0, in, (v4f *)in, nvec); }), 2.0 * bytes);
    // ---- 6. what makes the one-vector-per-thread copy fast: block size (XCD dealing granularity), one front or eight ----
#define ONE(BS, REMAP, label) report(label, time_ms([&] { hipLaunchKernelGGL((k_one<BS, REMAP>), dim3((unsigned)(nvec / BS)), dim3(BS), 0, 0, in, out); }), 2.0 * bytes)
    ONE(64, 0, "copy 1 vec/thread nt, 64-thread blocks");
    ONE(128, 0, "copy 1 vec/thread nt, 128-thread blocks");
    ONE(256, 0, "copy 1 vec/thread nt, 256-thread blocks");
    ONE(512, 0, "copy 1 vec/thread nt, 512-thread blocks");
    ONE(1024, 0, "copy 1 vec/thread nt, 1024-thread blocks");
    ONE(256, 1, "copy 1 vec/thread nt, 256-thread blocks, XCD remap (8 fronts)");
    ONE(1024, 1, "copy 1 vec/thread nt, 1024-thread blocks, XCD remap (8 fronts)");
#define WT(U, REMAP, label) report(label, time_ms([&] { hipLaunchKernelGGL((k_wavetile<U, REMAP>), dim3((unsigned)(nvec / (256 * U))), dim3(256), 0, 0, in, out); }), 2.0 * bytes)
    WT(1, 0, "copy wave tile 1 KiB, one tile per wave");
    WT(2, 0, "copy wave tile 2 KiB, one tile per wave");
    WT(4, 0, "copy wave tile 4 KiB, one tile per wave");
    WT(8, 0, "copy wave tile 8 KiB, one tile per wave");
    WT(4, 1, "copy wave tile 4 KiB, one tile per wave, XCD remap");
    WT(8, 1, "copy wave tile 8 KiB, one tile per wave, XCD remap");
    return 0;
}
