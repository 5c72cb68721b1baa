// tools/valubench2.hip -- what keeps v_pk_fma_f32 from issuing every 4 cycles in a real kernel?  Variants of a
// register-only loop (wave64, gfx950), 1 / 2 / 4 waves per SIMD, ns per VALU instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/valubench2 tools/valubench2.hip ; tools/valubench2 [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

struct Taps { f2 w[16]; };

#define FMA0(acc, w, x) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(w), "v"(x))
#define FMA1(acc, w, x) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(w), "v"(x))

// VALU instructions per inner iteration are returned through `per_iter` on the host side (see table in main)
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, Taps t)
{
    f2 p[16];
    for (int i = 0; i < 16; ++i) p[i] = f2{(float)threadIdx.x + i, (float)threadIdx.x + i + 1};
    const float x0 = out[threadIdx.x & 63];
    f2 xp = {x0, x0 + 1}, yp = {x0 + 2, x0 + 3};
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {                   // 128 independent, one SGPR pair
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) FMA0(p[i], t.w[0], xp);
        } else if constexpr (KIND == 1) {            // 128 independent, SGPR pair changes every instruction
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) FMA0(p[i], t.w[i], xp);
        } else if constexpr (KIND == 2) {            // op_sel alternates
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) { if (i & 1) FMA1(p[i], t.w[i >> 1], xp); else FMA0(p[i], t.w[i >> 1], xp); }
        } else if constexpr (KIND == 3) {            // one dependent chain
#pragma unroll
            for (int rep = 0; rep < 128; ++rep) FMA0(p[0], t.w[rep & 15], xp);
        } else if constexpr (KIND == 4) {            // two chains
#pragma unroll
            for (int rep = 0; rep < 64; ++rep) { FMA0(p[0], t.w[rep & 15], xp); FMA0(p[1], t.w[rep & 15], xp); }
        } else if constexpr (KIND == 5) {            // four chains
#pragma unroll
            for (int rep = 0; rep < 32; ++rep) { FMA0(p[0], t.w[rep & 15], xp); FMA0(p[1], t.w[rep & 15], xp); FMA0(p[2], t.w[rep & 15], xp); FMA0(p[3], t.w[rep & 15], xp); }
        } else if constexpr (KIND == 6) {            // 16 fma + 2 pk_mov per group (8 groups): the 1-D kernel's mix
#pragma unroll
            for (int rep = 0; rep < 8; ++rep) {
                asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(yp) : "v"(xp), "v"(p[15]));
#pragma unroll
                for (int i = 0; i < 8; ++i) FMA0(p[i], t.w[rep], yp);
                asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(xp) : "v"(yp), "v"(p[14]));
#pragma unroll
                for (int i = 8; i < 16; ++i) FMA1(p[i], t.w[rep], xp);
            }
        } else if constexpr (KIND == 7) {            // an s_nop after every 4th fma
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) { FMA0(p[i], t.w[i], xp); if ((i & 3) == 3) asm volatile("s_nop 0"); }
        } else if constexpr (KIND == 8) {            // a SALU op after every 4th fma
            int sc = it;
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) { FMA0(p[i], t.w[i], xp); if ((i & 3) == 3) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sc)); }
            if (sc == -12345) out[1] = 1.0f;
        } else if constexpr (KIND == 9) {            // 3-source form: fold-like, new destination each time (dst != src2)
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(p[i]) : "s"(t.w[i]), "v"(xp), "v"(p[(i + 5) & 15]));
        } else if constexpr (KIND == 10) {           // v_pk_mul + v_pk_add pairs (streaming kernels), 8 chains
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    f2 q;
                    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(q) : "s"(t.w[i]), "v"(xp));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q));
                }
        }
    }
    float s = xp.x + yp.y;
    for (int i = 0; i < 16; ++i) s += p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
void run(const char *name, float *buf, int iters)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    Taps t;
    for (int i = 0; i < 16; ++i) t.w[i] = f2{1.0f + i * 1e-3f, 1.0f - i * 1e-3f};
    const double valu_per_iter = KIND == 6 ? 144.0 : 128.0;
    for (int wps : {1, 2, 4}) {
        const int grid = 256 * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, buf, 10, t);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, buf, iters, t);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-44s waves/SIMD=%d : %8.3f ms  -> %.2f ns per VALU instruction per SIMD\n", name, wps, ms, ms * 1e6 / ((double)iters * valu_per_iter * wps));
    }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float *buf; CK(hipMalloc(&buf, 4096)); CK(hipMemset(buf, 0, 4096));
    run<0>("independent, one SGPR pair", buf, iters);
    run<1>("independent, SGPR pair changes every instr", buf, iters);
    run<2>("independent, op_sel alternates", buf, iters);
    run<3>("1 dependent chain", buf, iters);
    run<4>("2 chains", buf, iters);
    run<5>("4 chains", buf, iters);
    run<6>("8 fma + 1 pk_mov groups (1-D kernel mix)", buf, iters);
    run<7>("s_nop after every 4th", buf, iters);
    run<8>("s_add after every 4th", buf, iters);
    run<9>("3-source form, dst != src2", buf, iters);
    run<10>("pk_mul + pk_add, 8 chains", buf, iters);
    return 0;
}
