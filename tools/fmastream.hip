// tools/fmastream.hip -- the power wall: a kernel that does NOTHING but stream 16 GiB in + 16 GiB out (the headline
// workload's bytes) and issue K packed FMAs per 16-byte vector on registers (the headline kernel issues 130 per
// vector at n=32).  No halo, no window, no LDS unless asked: whatever time this takes is the floor for any kernel
// with that many v_pk_fma_f32 per byte on this part.
//   hipcc --offload-arch=gfx950 -O3 -o tools/fmastream tools/fmastream.hip ; tools/fmastream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

struct Taps { f2 w[16]; };

// one wave-iteration: 8 vectors per lane (a 2048-sample tile per wave, like the headline kernel), K*8 FMAs
template <int K, int LDS, int NT = 1>
__global__ __launch_bounds__(256, 4) void k(const f4 *__restrict__ in, f4 *__restrict__ out, size_t nvec, Taps t)
{
    __shared__ f4 slab[LDS ? 4 * 64 * 9 : 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t nwaves = (size_t)gridDim.x * 4, wave = (size_t)blockIdx.x * 4 + wv;
    for (size_t tile = wave; tile * 512 < nvec; tile += nwaves) {
        const f4 *src = in + tile * 512;
        f4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = NT ? __builtin_bit_cast(f4, __builtin_nontemporal_load(reinterpret_cast<const u4 *>(src + j * 64 + lane))) : src[j * 64 + lane];
        if constexpr (LDS) {                         // the staging round trip of the real kernel: coalesced rows in, per-lane rows out
            f4 *mine = slab + wv * 64 * 9;
#pragma unroll
            for (int j = 0; j < 8; ++j) mine[(j * 64 + lane) + (j * 64 + lane) / 8] = v[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = mine[lane * 9 + j];
            if constexpr (LDS > 1) {                 // plus the window's halo: 16 more vectors read per lane (3x amplification)
                f4 h = v[0];
#pragma unroll
                for (int j = 0; j < 16; ++j) { const f4 q = mine[((lane + 1 + j / 8) & 63) * 9 + (j & 7)]; h += q; }
                v[0] = h;
            }
        }
        f2 acc[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[2 * j] = f2{v[j].x, v[j].y}; acc[2 * j + 1] = f2{v[j].z, v[j].w}; }
        const f2 x = acc[3];
#pragma unroll
        for (int r = 0; r < K / 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "s"(t.w[(r + i) & 15]), "v"(x));
        f4 *dst = out + tile * 512;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f4 o = f4{acc[2 * j].x, acc[2 * j].y, acc[2 * j + 1].x, acc[2 * j + 1].y};
            if (NT) __builtin_nontemporal_store(__builtin_bit_cast(u4, o), reinterpret_cast<u4 *>(dst + j * 64 + lane));
            else dst[j * 64 + lane] = o;
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

template <int K, int LDS, int NT = 1>
void run(const f4 *in, f4 *out, size_t nvec)
{
    Taps t;
    for (int i = 0; i < 16; ++i) t.w[i] = f2{1e-3f * i, -1e-3f * i};
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<K, LDS, NT>), dim3(1024), dim3(256), 0, 0, in, out, nvec, t);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<K, LDS, NT>), dim3(1024), dim3(256), 0, 0, in, out, nvec, t);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    printf("pk_fma per 16-B vector = %3d, LDS mode %d, %s : %7.3f ms  -> %6.0f GB/s in+out (%4.1f %% of 8 TB/s)\n", K, LDS, NT ? "nontemporal" : "plain      ", ms,
           2.0 * nvec * 16 / ms / 1e6, 2.0 * nvec * 16 / ms / 1e6 / 80.0);
}

int main(int argc, char **argv)
{
    const bool zeros = argc > 1 && argv[1][0] == 'z';          // all-zero input: how much of the time is data-dependent power?
    const size_t nvec = (size_t)4096 * (1 << 20) / 4;       // 16 GiB of fp32
    f4 *in, *out;
    CK(hipMalloc(&in, nvec * 16)); CK(hipMalloc(&out, nvec * 16));
    if (zeros) CK(hipMemset(in, 0, nvec * 16));
    else hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<float *>(in), nvec * 4);
    CK(hipDeviceSynchronize());
    run<0, 0>(in, out, nvec);
    run<32, 0>(in, out, nvec);
    run<64, 0>(in, out, nvec);
    run<96, 0>(in, out, nvec);
    run<128, 0>(in, out, nvec);      // ~ the headline kernel's 130
    run<160, 0>(in, out, nvec);
    run<128, 1>(in, out, nvec);
    run<128, 2>(in, out, nvec);
    run<64, 2>(in, out, nvec);       // ~ n=16
    run<0, 0, 0>(in, out, nvec);     // plain loads / stores instead of nontemporal
    run<128, 0, 0>(in, out, nvec);
    run<128, 2, 0>(in, out, nvec);
    return 0;
}
