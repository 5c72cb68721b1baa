// tools/membench.hip -- HBM access-pattern microbenchmarks behind the tile geometry of the 1-D kernel.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/membench tools/membench.hip && /tmp/membench
// Each kernel reads (and optionally writes) a 16 GiB fp32 buffer; prints GB/s.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// K0: grid-stride float4 read (+ optional write): the "copy kernel" shape
__global__ __launch_bounds__(256) void k_stream(const float4 *__restrict__ in, float4 *__restrict__ out, size_t nvec, int write)
{
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = in[i];
        if (write) out[i] = v; else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
    if (!write && acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc;
}

// K1: persistent waves, one 8 KiB tile (8 x 1 KiB wave-instructions) per wave per iteration,
// next tile = + nwaves (the 1-D kernel's shape).  unroll = loads issued before the first use.
template <int TILE_VECS_PER_LANE>
__global__ __launch_bounds__(256) void k_tiles(const float4 *__restrict__ in, float4 *__restrict__ out, unsigned ntiles,
                                               int write, int xcd_remap, int prefetch)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nblk = gridDim.x;
    const unsigned blk = xcd_remap ? (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const unsigned nwaves = nblk * 4;
    float4 acc = make_float4(0, 0, 0, 0);
    float4 v[TILE_VECS_PER_LANE], nx[TILE_VECS_PER_LANE];
    unsigned tile = blk * 4 + wave;
    if (prefetch && tile < ntiles) {
        const float4 *src = in + (size_t)tile * (64 * TILE_VECS_PER_LANE);
#pragma unroll
        for (int s = 0; s < TILE_VECS_PER_LANE; ++s) nx[s] = src[lane + 64 * s];
    }
    for (; tile < ntiles; tile += nwaves) {
        const float4 *src = in + (size_t)tile * (64 * TILE_VECS_PER_LANE);
        if (prefetch) {
#pragma unroll
            for (int s = 0; s < TILE_VECS_PER_LANE; ++s) v[s] = nx[s];
            if (tile + nwaves < ntiles) {
                const float4 *s2 = in + (size_t)(tile + nwaves) * (64 * TILE_VECS_PER_LANE);
#pragma unroll
                for (int s = 0; s < TILE_VECS_PER_LANE; ++s) nx[s] = s2[lane + 64 * s];
            }
        } else {
#pragma unroll
            for (int s = 0; s < TILE_VECS_PER_LANE; ++s) v[s] = src[lane + 64 * s];
        }
        if (write) {
            float4 *dst = out + (size_t)tile * (64 * TILE_VECS_PER_LANE);
#pragma unroll
            for (int s = 0; s < TILE_VECS_PER_LANE; ++s) dst[lane + 64 * s] = v[s];
        } else {
#pragma unroll
            for (int s = 0; s < TILE_VECS_PER_LANE; ++s) { acc.x += v[s].x; acc.y += v[s].y; acc.z += v[s].z; acc.w += v[s].w; }
        }
    }
    if (!write && acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc;
}

template <typename F>
static double time_ms(F launch, int iters = 5)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b));
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main(int argc, char **argv)
{
    const size_t bytes = (size_t)16 << 30;
    float4 *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    const size_t nvec = bytes / 16;
    const unsigned ntiles8k = (unsigned)(bytes / 8192);

    for (int write = 0; write <= 1; ++write) {
        for (int g : {1024, 2048, 4096, 8192, 65536}) {
            double ms = time_ms([&] { hipLaunchKernelGGL(k_stream, dim3(g), dim3(256), 0, 0, in, out, nvec, write); });
            printf("k_stream  write=%d grid=%6d : %7.3f ms  read %7.1f GB/s  total %7.1f GB/s\n", write, g, ms,
                   bytes / ms / 1e6, (write ? 2.0 : 1.0) * bytes / ms / 1e6);
        }
    }
    for (int write = 0; write <= 1; ++write)
        for (int pf = 0; pf <= 1; ++pf)
            for (int remap = 0; remap <= 1; ++remap)
                for (int g : {1024, 2048, 4096, 16384}) {
                    double ms = time_ms([&] { hipLaunchKernelGGL(k_tiles<8>, dim3(g), dim3(256), 0, 0, in, out, ntiles8k, write, remap, pf); });
                    printf("k_tiles8K write=%d prefetch=%d remap=%d grid=%6d : %7.3f ms  read %7.1f GB/s  total %7.1f GB/s\n", write, pf,
                           remap, g, ms, bytes / ms / 1e6, (write ? 2.0 : 1.0) * bytes / ms / 1e6);
                }
    for (int g : {1024, 4096}) {
        double ms = time_ms([&] { hipLaunchKernelGGL(k_tiles<4>, dim3(g), dim3(256), 0, 0, in, out, ntiles8k * 2, 0, 1, 1); });
        printf("k_tiles4K write=0 prefetch=1 remap=1 grid=%6d : %7.3f ms  read %7.1f GB/s\n", g, ms, bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(k_tiles<16>, dim3(g), dim3(256), 0, 0, in, out, ntiles8k / 2, 0, 1, 1); });
        printf("k_tiles16K write=0 prefetch=1 remap=1 grid=%6d : %7.3f ms  read %7.1f GB/s\n", g, ms, bytes / ms / 1e6);
    }
    return 0;
}
