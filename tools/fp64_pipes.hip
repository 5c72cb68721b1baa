// tools/fp64_pipes.hip -- round 3: does gfx950 run v_mfma_f64_16x16x4_f64 beside v_fma_f64, or do the two share the fp64 datapath?
// The fp64 1-D kernel (config 5) is bound by vector issue (65 v_fma_f64 per output); if the matrix pipe is separate, part of
// the outputs of a tile could be computed as a banded Toeplitz product there while the vector unit does the rest.
//   hipcc --offload-arch=gfx950 -O3 -o tools/fp64_pipes tools/fp64_pipes.hip && tools/fp64_pipes
// Prints, per kernel, the time and the fp64 multiply-adds per cycle per SIMD at the clock the event timing implies.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

// VF vector multiply-adds and MF matrix instructions per loop trip, all independent chains
template <int VF, int MF>
__global__ __launch_bounds__(256) void k_mix(double *out, int trips, double seed)
{
    double acc[VF > 0 ? VF : 1];
    v4d d[MF > 0 ? MF : 1];
    const double x = seed + threadIdx.x * 1e-9, w = 1.0 + seed * 1e-9;
#pragma unroll
    for (int i = 0; i < (VF > 0 ? VF : 1); ++i) acc[i] = i * seed;
#pragma unroll
    for (int i = 0; i < (MF > 0 ? MF : 1); ++i) d[i] = v4d{seed, 0.0, seed, 0.0};
    for (int t = 0; t < trips; ++t) {
        // interleave: one matrix instruction, then its share of the vector ones
#pragma unroll
        for (int m = 0; m < (MF > 0 ? MF : 1); ++m) {
            if constexpr (MF > 0) d[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, w, d[m], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < VF; ++i)
                if (i % (MF > 0 ? MF : 1) == m) acc[i] = __builtin_fma(acc[i], w, x);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < VF; ++i) s += acc[i];
#pragma unroll
    for (int i = 0; i < MF; ++i) s += d[i].x + d[i].y + d[i].z + d[i].w;
    if (s == 12345.678) out[blockIdx.x * 256 + threadIdx.x] = s;     // never true: keeps the chains alive
}

template <int VF, int MF>
static void run(const char *what, double *d_out, int blocks, int trips, int waves_per_simd)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_mix<VF, MF>), dim3(blocks), dim3(256), 0, 0, d_out, trips / 10, 0.5);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_mix<VF, MF>), dim3(blocks), dim3(256), 0, 0, d_out, trips, 0.5);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    // per SIMD: waves_per_simd waves, each trips x (VF x 64 vector + MF x 1024 matrix) multiply-adds
    const double vfma = (double)waves_per_simd * trips * VF * 64.0, mfma = (double)waves_per_simd * trips * MF * 1024.0;
    const double ns = best * 1e6;
    printf("%-44s VF=%2d MF=%d waves/SIMD=%d: %8.3f ms   vector %6.2f + matrix %6.2f = %6.2f multiply-adds per ns per SIMD  (chip: %.1f TFLOP/s)\n", what, VF, MF,
           waves_per_simd, best, vfma / ns, mfma / ns, (vfma + mfma) / ns, 2.0 * (vfma + mfma) / ns * 1024.0 * 1e-3);
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs, %d MHz\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    double *d_out;
    CK(hipMalloc(&d_out, sizeof(double) * 256 * cus * 8));
    const int trips = 20000;
    for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = cus * wps;                                   // 256 threads = one wave per SIMD per block
        run<16, 0>("vector only", d_out, blocks, trips, wps);
        run<0, 4>("matrix only", d_out, blocks, trips, wps);
        run<16, 4>("both, 4 vector per matrix instruction", d_out, blocks, trips, wps);
        run<32, 4>("both, 8 vector per matrix instruction", d_out, blocks, trips, wps);
        run<48, 4>("both, 12 vector per matrix instruction", d_out, blocks, trips, wps);
        run<64, 4>("both, 16 vector per matrix instruction", d_out, blocks, trips, wps);
    }
    return 0;
}
