// tools/fmastream64.hip -- fp64 twin of fmastream2: does the energy of v_fma_f64 set the time of the fp64 n=32 kernel, and would
// replacing half of the FMAs by adds (folding the symmetric taps: 32 adds + 33 FMAs instead of 65 FMAs per output) buy anything?
// Streams 16 GiB in + 16 GiB out, one 8 KiB tile per wave in block order, MODE 0: K v_fma_f64 per double, MODE 1: K/2 v_add_f64 +
// K/2 v_fma_f64 per double.   hipcc --offload-arch=gfx950 -O3 -o tools/fmastream64 tools/fmastream64.hip ; tools/fmastream64
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
struct Taps { double w[16]; };

template <int K, int MODE>
__global__ __launch_bounds__(256, 3) void k(const d2 *__restrict__ in, d2 *__restrict__ out, unsigned ntiles, Taps t)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned nb8 = gridDim.x >> 3;
    unsigned blk = blockIdx.x;
    if (blk < nb8 * 8u) blk = (blk & 7u) * nb8 + (blk >> 3);
    const unsigned tile = blk * 4 + wv;
    if (tile >= ntiles) return;
    const d2 *src = in + (size_t)tile * 512;
    d2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __builtin_bit_cast(d2, __builtin_nontemporal_load(reinterpret_cast<const u4 *>(src + j * 64 + lane)));
    double acc[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[2 * j] = v[j].x; acc[2 * j + 1] = v[j].y; }
    const double x = acc[3], y = acc[5];
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 1 && (r & 1)) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
            else asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(t.w[(r + i) & 15]), "v"(x));
        }
    d2 *dst = out + (size_t)tile * 512;
#pragma unroll
    for (int j = 0; j < 8; ++j) __builtin_nontemporal_store(__builtin_bit_cast(u4, d2{acc[2 * j], acc[2 * j + 1]}), reinterpret_cast<u4 *>(dst + j * 64 + lane));
}
__global__ void fill(double *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull + 0x5A17601Aull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
}
template <int K, int MODE> void run(const d2 *in, d2 *out, size_t nvec)
{
    Taps t; for (int i = 0; i < 16; ++i) t.w[i] = 1e-3 * (i - 7);
    const unsigned ntiles = (unsigned)(nvec / 512), grid = ((ntiles + 3) / 4 + 7) & ~7u;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto launch = [&] { hipLaunchKernelGGL((k<K, MODE>), dim3(grid), dim3(256), 0, 0, in, out, ntiles, t); };
    launch(); CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m); }
    std::sort(ms.begin(), ms.end());
    printf("fp64 ops per double = %3d, %s : %7.3f ms -> %6.0f GB/s in+out (%4.1f %%)\n", K, MODE ? "half adds, half FMAs" : "all FMAs           ", ms[2],
           2.0 * nvec * 16 / ms[2] / 1e6, 2.0 * nvec * 16 / ms[2] / 1e6 / 80.0);
}
int main()
{
    const size_t nvec = (size_t)1 << 30;                      // 16 GiB
    d2 *in, *out; CK(hipMalloc(&in, nvec * 16)); CK(hipMalloc(&out, nvec * 16));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<double *>(in), nvec * 2); CK(hipDeviceSynchronize());
    run<0, 0>(in, out, nvec); run<32, 0>(in, out, nvec); run<64, 0>(in, out, nvec); run<64, 1>(in, out, nvec); run<48, 0>(in, out, nvec); run<40, 0>(in, out, nvec);
    return 0;
}
