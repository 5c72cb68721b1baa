// tools/fmastream2.hip -- which WORK DISTRIBUTION lets a tile kernel (8 KiB or 4 KiB of output per wave, K packed FMAs per
// 16-byte vector, optional LDS round trip: the shape of sg1d_center_kernel) stream 16 GiB in + 16 GiB out fastest.
// tools/membench2 showed a one-vector-per-thread, non-persistent copy at 6.5 TB/s against 5.2 TB/s for persistent
// 8 KiB tiles; this program separates "persistent vs dispatched in order" from "bytes per wave".
//   SCHED 0  persistent, static stride (tile += nwaves), no prefetch            (tools/fmastream.hip)
//   SCHED 1  one tile per wave, grid = ntiles/4, blocks dispatched in order
//   SCHED 2  persistent, tiles handed out in order by an atomic counter (index fetched one tile ahead)
//   SCHED 3  as 1 with the XCD remap (each XCD sweeps its own eighth of the buffer: 8 fronts)
//   SCHED 4  persistent, static stride, XCD remap, next tile's DATA prefetched into registers (sg1d_center_kernel r01)
//   SCHED 5  as 2 with the next tile's data prefetched into registers
//   hipcc --offload-arch=gfx950 -O3 -o tools/fmastream2 tools/fmastream2.hip ; tools/fmastream2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

struct Taps { f2 w[16]; };

__device__ __forceinline__ f4 ldnt(const f4 *p) { return __builtin_bit_cast(f4, __builtin_nontemporal_load(reinterpret_cast<const u4 *>(p))); }
__device__ __forceinline__ void stnt(f4 *p, f4 v) { __builtin_nontemporal_store(__builtin_bit_cast(u4, v), reinterpret_cast<u4 *>(p)); }

template <int K, int LDS, int U>
__device__ __forceinline__ void body(f4 (&v)[U], f4 *mine, int lane, const Taps &t)
{
    if constexpr (LDS) {                         // the staging round trip of the real kernel: coalesced rows in, per-lane rows out
#pragma unroll
        for (int j = 0; j < U; ++j) mine[(j * 64 + lane) + (j * 64 + lane) / U] = v[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = mine[lane * (U + 1) + j];
        if constexpr (LDS > 1) {                 // plus the window's halo: 16 more vectors read per lane
            f4 h = v[0];
#pragma unroll
            for (int j = 0; j < 16; ++j) { const f4 q = mine[((lane + 1 + j / U) & 63) * (U + 1) + (j % U)]; h += q; }
            v[0] = h;
        }
    }
    f2 acc[2 * U];
#pragma unroll
    for (int j = 0; j < U; ++j) { acc[2 * j] = f2{v[j].x, v[j].y}; acc[2 * j + 1] = f2{v[j].z, v[j].w}; }
    const f2 x = acc[3];
#pragma unroll
    for (int r = 0; r < K / 2; ++r)
#pragma unroll
        for (int i = 0; i < 2 * U; ++i)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "s"(t.w[(r + i) & 15]), "v"(x));
#pragma unroll
    for (int j = 0; j < U; ++j) v[j] = f4{acc[2 * j].x, acc[2 * j].y, acc[2 * j + 1].x, acc[2 * j + 1].y};
}

template <int K, int LDS, int SCHED, int U>
__global__ __launch_bounds__(256, 4) void k(const f4 *__restrict__ in, f4 *__restrict__ out, unsigned ntiles, Taps t, unsigned *counter)
{
    __shared__ f4 slab[LDS ? 4 * 64 * (U + 1) : 1];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f4 *mine = slab + (LDS ? wv * 64 * (U + 1) : 0);
    constexpr int TV = 64 * U;
    f4 v[U];
    auto load = [&](f4 (&d)[U], unsigned tile) {
        const f4 *src = in + (size_t)tile * TV;
#pragma unroll
        for (int j = 0; j < U; ++j) d[j] = ldnt(src + j * 64 + lane);
    };
    auto store = [&](unsigned tile) {
        f4 *dst = out + (size_t)tile * TV;
#pragma unroll
        for (int j = 0; j < U; ++j) stnt(dst + j * 64 + lane, v[j]);
    };
    auto grab = [&]() -> unsigned {
        unsigned r = 0;
        if (lane == 0) r = atomicAdd(counter, 1u);
        return __builtin_amdgcn_readfirstlane(r);
    };
    if constexpr (SCHED == 0) {
        const unsigned nwaves = gridDim.x * 4;
        for (unsigned tile = blockIdx.x * 4 + wv; tile < ntiles; tile += nwaves) { load(v, tile); body<K, LDS, U>(v, mine, lane, t); store(tile); }
    } else if constexpr (SCHED == 1 || SCHED == 3) {
        const unsigned nblk = gridDim.x;
        const unsigned blk = SCHED == 3 ? (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3) : blockIdx.x;
        const unsigned tile = blk * 4 + wv;
        if (tile < ntiles) { load(v, tile); body<K, LDS, U>(v, mine, lane, t); store(tile); }
    } else if constexpr (SCHED == 2) {
        unsigned tile = grab();
        while (tile < ntiles) {
            const unsigned next = grab();
            load(v, tile); body<K, LDS, U>(v, mine, lane, t); store(tile);
            tile = next;
        }
    } else if constexpr (SCHED == 4) {
        const unsigned nblk = gridDim.x;
        const unsigned blk = (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3);
        const unsigned nwaves = nblk * 4;
        f4 nx[U];
        unsigned tile = blk * 4 + wv;
        if (tile < ntiles) load(nx, tile);
        for (; tile < ntiles; tile += nwaves) {
#pragma unroll
            for (int j = 0; j < U; ++j) v[j] = nx[j];
            if (tile + nwaves < ntiles) load(nx, tile + nwaves);
            body<K, LDS, U>(v, mine, lane, t); store(tile);
        }
    } else {
        f4 nx[U];
        unsigned tile = grab();
        if (tile < ntiles) load(nx, tile);
        while (tile < ntiles) {
            const unsigned next = grab();
#pragma unroll
            for (int j = 0; j < U; ++j) v[j] = nx[j];
            if (next < ntiles) load(nx, next);
            body<K, LDS, U>(v, mine, lane, t); store(tile);
            tile = next;
        }
    }
}

__global__ void fill(float *p, size_t n)          // noisy data: realistic bit toggling (power) in the FMAs and on the wires
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull + 0x5A17601Aull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = (float)(z >> 40) * (1.0f / 16777216.0f) - 0.5f;
    }
}

static unsigned *d_counter;

template <int K, int LDS, int SCHED, int U>
void run(const f4 *in, f4 *out, size_t nvec, unsigned persistent_grid = 1024)
{
    Taps t;
    for (int i = 0; i < 16; ++i) t.w[i] = f2{1e-3f * i, -1e-3f * i};
    const unsigned ntiles = (unsigned)(nvec / (64 * U));
    const bool persistent = SCHED == 0 || SCHED == 2 || SCHED == 4 || SCHED == 5;
    const unsigned grid = persistent ? persistent_grid : (ntiles + 3) / 4;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto launch = [&] {
        if (SCHED == 2 || SCHED == 5) CK(hipMemsetAsync(d_counter, 0, 4, 0));
        hipLaunchKernelGGL((k<K, LDS, SCHED, U>), dim3(grid), dim3(256), 0, 0, in, out, ntiles, t, d_counter);
    };
    launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    printf("K=%3d LDS=%d SCHED=%d U=%d grid=%7u : %7.3f ms -> %6.0f GB/s in+out (%4.1f %%)\n", K, LDS, SCHED, U, grid, ms[2],
           2.0 * nvec * 16 / ms[2] / 1e6, 2.0 * nvec * 16 / ms[2] / 1e6 / 80.0);
    fflush(stdout);
}

template <int K, int LDS, int U>
void sweep(const f4 *in, f4 *out, size_t nvec)
{
    run<K, LDS, 0, U>(in, out, nvec);
    run<K, LDS, 1, U>(in, out, nvec);
    run<K, LDS, 2, U>(in, out, nvec);
    run<K, LDS, 3, U>(in, out, nvec);
    run<K, LDS, 4, U>(in, out, nvec);
    run<K, LDS, 5, U>(in, out, nvec);
}

int main()
{
    const size_t nvec = (size_t)4096 * (1 << 20) / 4;       // 16 GiB of fp32
    f4 *in, *out;
    CK(hipMalloc(&in, nvec * 16)); CK(hipMalloc(&out, nvec * 16)); CK(hipMalloc(&d_counter, 4));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<float *>(in), nvec * 4);
    CK(hipDeviceSynchronize());
    sweep<0, 0, 8>(in, out, nvec);
    sweep<0, 0, 4>(in, out, nvec);
    sweep<0, 2, 8>(in, out, nvec);
    sweep<128, 0, 8>(in, out, nvec);
    sweep<128, 2, 8>(in, out, nvec);      // ~ the headline kernel
    sweep<128, 2, 4>(in, out, nvec);
    sweep<64, 2, 8>(in, out, nvec);       // ~ n = 16
    // persistent shapes at other grid sizes
    run<128, 2, 2, 8>(in, out, nvec, 512);
    run<128, 2, 2, 8>(in, out, nvec, 768);
    run<128, 2, 5, 8>(in, out, nvec, 512);
    run<128, 2, 5, 8>(in, out, nvec, 768);
    run<128, 2, 4, 8>(in, out, nvec, 768);
    return 0;
}
