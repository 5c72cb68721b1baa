// tools/valubench.hip -- VALU issue-rate microbenchmark (wave64, gfx950): cycles per instruction per SIMD for
// v_fma_f32, v_pk_fma_f32, v_fma_f64 at 1..8 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o tools/valubench tools/valubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float w)
{
    float a[16]; f2 p[16]; double d[16];
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x + i; p[i] = f2{a[i], a[i] + 1}; d[i] = a[i]; }
    const float x = out[threadIdx.x & 63];
    const f2 xp = {x, x + 1};
    const double xd = x, wd = w;
    const f2 wp = {w, w};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(w), "v"(x));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "s"(wp), "v"(xp));
                if (KIND == 2) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[i]) : "v"(wd), "v"(xd));
                if (KIND == 3) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "s"(wd), "v"(xd));
                if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(wp), "v"(xp));
            }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
void run(const char *name, float *buf, int iters)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int wps : {1, 2, 4, 8}) {                        // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
        const int grid = 256 * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, buf, 10, 1.0f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, buf, iters, 1.0f);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double inst_per_simd = (double)iters * 128 * wps;          // each wave: iters*128 instructions
        printf("%-28s waves/SIMD=%d : %8.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles @2.4GHz)\n", name, wps, ms,
               ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;          // 100000+ = long enough to see the sustained (power-managed) clock
    float *buf; CK(hipMalloc(&buf, 4096)); CK(hipMemset(buf, 0, 4096));
    run<0>("v_fmac_f32 (sgpr tap)", buf, iters);
    run<1>("v_pk_fma_f32 (sgpr pair)", buf, iters);
    run<4>("v_pk_fma_f32 (vgpr pair)", buf, iters);
    run<2>("v_fmac_f64 (vgpr tap)", buf, iters);
    run<3>("v_fma_f64 (sgpr tap)", buf, iters);
    return 0;
}
