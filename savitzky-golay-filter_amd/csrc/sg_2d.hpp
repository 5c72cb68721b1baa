// sg_2d.hpp -- shared by the two 2-D kernel files.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>

#include "sg_internal.h"
#include "sg_pk.hpp"

namespace sg {

struct Job2D {
    const float *in;
    float       *out;
    int rows, cols, in_stride, out_stride;
    long long in_pitch, out_pitch;      // elements between images
    int nx, ny;
    int boundary;                       // Savgol2DBoundary (anything else was mapped to CONSTANT by the host)
    float scale;
    int tiles_x, tiles_y;
    int accumulate;                     // rolling kernel only: out += result (the second pass of a kernel split over two launches)
};

// frame coordinate fix-up of the padded modes (reference src/savgol2d.c:428-445); VALID only clamps
// (its out-of-frame reads feed outputs that are never stored)
__device__ __forceinline__ int fix_index(int i, int n, int boundary)
{
    if (boundary == SAVGOL2D_BOUNDARY_REFLECT) {
        if (i < 0) i = -i - 1; else if (i >= n) i = 2 * n - i - 1;
    }
    if (i < 0) i = 0; else if (i >= n) i = n - 1;
    return i;
}

// ---- shared by the rolling kernels (sg_2d_roll.hip, sg_2d_hf.hip) ----
// a + s * b, s = {+-1, +-1} in an SGPR pair.  Deliberately NOT inline asm: the hazard recogniser counts no wait
// states for inline asm, so a chain of asm multiply-adds gets an s_nop per step unless compiler-visible
// instructions (these folds) sit between a result and its use.
__device__ __forceinline__ f32x2 pk_fold(const f32x2 s, const f32x2 b, const f32x2 a)
{
    return __builtin_elementwise_fma(s, b, a);
}
// (q.y, q.z) of ONE 16-byte LDS read: the middle pair of a quad sits in an odd-aligned register pair, and the compiler copies it
// out with two v_mov_b32 (it only uses v_pk_mov_b32 for a pair that straddles two reads).  One v_pk_mov_b32 does it: 10 VALU
// instructions fewer per row at n = 7, rank 2 (135 -> 125).
__device__ __forceinline__ f32x2 pk_middle(const f32x2 lo, const f32x2 hi)
{
#ifdef SG_ROLL_PLAIN_MIDDLE
    return pk_straddle(lo, hi);
#else
    f32x2 r;
    asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
#endif
}
// fix_index (sg_2d.hpp) without branches: the row index is wave-uniform, so this is a handful of SALU selects
__device__ __forceinline__ int fix_row(int i, int n, bool reflect)
{
    const int below = reflect ? ~i : 0;                      // i < 0:  -i-1 | 0
    const int above = reflect ? 2 * n - 1 - i : n - 1;       // i >= n
    int a = i < 0 ? below : (i >= n ? above : i);
    a = a < 0 ? 0 : a;                                       // frames smaller than the window: one reflection, then clamp
    return a >= n ? n - 1 : a;
}

// factors: per term Q_t[0..2N], pad, G_t[0..2N], pad (sg2d_factors_from_kernel).  Returns false when a vector has no
// definite parity (cannot happen for a least-squares kernel on a symmetric window; arbitrary kernels may).
inline bool vector_parity(const float *v, int n, float *sign)
{
    float vmax = 0.0f;
    for (int k = 0; k <= 2 * n; ++k) vmax = fmaxf(vmax, fabsf(v[k]));
    const float tol = 4e-6f * vmax;
    bool even = true, odd = true;
    for (int k = 0; k < n; ++k) {
        if (fabsf(v[k] - v[2 * n - k]) > tol) even = false;
        if (fabsf(v[k] + v[2 * n - k]) > tol) odd = false;
    }
    if (fabsf(v[n]) > tol) odd = false;
    if (even) { *sign = 1.0f; return true; }
    if (odd) { *sign = -1.0f; return true; }
    return false;
}

constexpr int SEP_MAX_TERMS = 4;        // per output: rank of a bivariate polynomial of total degree <= 6 is at most 4 (parity in y)
constexpr int SEP_MAX_OUTPUTS = 3;      // gradient = 2, Hessian = 3 outputs computed from ONE read of the input tile

// what the separable kernel produces from each input tile: `outputs` frames, output o = scale[o] * sum of its terms
struct SepPlan {
    int    outputs;
    int    terms[SEP_MAX_OUTPUTS];      // factors are stored output after output, term after term
    float  scale[SEP_MAX_OUTPUTS];
    float *out[SEP_MAX_OUTPUTS];        // same row stride / image pitch for all
    int    transposed;                  // 1: vertical pass first (every output y-dominant: deriv_y >= 1, deriv_y > deriv_x), see sg_2d_sep.hip
    int    centre;                      // 1: the first pass runs on centred samples (s - c) and adds c * first_sum[t] back (derivative kernels; sg_2d_hf.hip, R6.8)
    float  first_sum[SEP_MAX_TERMS];    // what term t's first-pass factor sums to in the reference's dense table (unscaled), single-output plans
};
inline bool sg2d_y_dominant(int deriv_x, int deriv_y) { return deriv_y >= 1 && deriv_y > deriv_x; }

// Row bands of the strip kernels (sg_2d_roll.hip, sg_2d_dense.hip): an item = one strip x one band, `nwaves` resident
// waves take items round robin.  Every band pays 2n warm-up rows (weighted `warm`: they cost full work in the dense
// kernel, loads only in the separable one), and the last round of items may leave waves idle; pick the band count with
// the smallest cost per wave.
inline unsigned choose_bands(int rows, unsigned long long images_x_strips, unsigned nwaves, int n, double warm)
{
    unsigned best = 1;
    double best_cost = 1e300;
    for (unsigned b = 1; b <= 64u && (int)b <= rows; ++b) {
        const int band_rows = (rows + (int)b - 1) / (int)b;
        const unsigned real_b = (unsigned)((rows + band_rows - 1) / band_rows);
        if (real_b != b) continue;
        if (b > 1 && band_rows < 4 * n) break;
        const unsigned long long items = images_x_strips * b;
        const unsigned long long rounds = (items + nwaves - 1) / nwaves;
        const double cost = (double)rounds * ((double)band_rows + warm * 2.0 * n);
        if (cost < best_cost * 0.999) { best_cost = cost; best = b; }
    }
    return best;
}

// sg_2d_roll.hip: rolling-window kernel, half windows 1..16.  The file is compiled once per
// half-window group (SEP_ROLL_MIN_N..SEP_ROLL_MAX_N under the name SEP_ROLL_FN, see the Makefile) so the groups
// build in parallel; each returns 0 = launched, 1 = not covered (other group, or the caller uses the tile kernel).
int sg2d_launch_rolling_g0(int n, int terms, const Job2D &job, const float *factors, float scale, unsigned images, int cu_count, hipStream_t st);
int sg2d_launch_rolling_g1(int n, int terms, const Job2D &job, const float *factors, float scale, unsigned images, int cu_count, hipStream_t st);
int sg2d_launch_rolling_g2(int n, int terms, const Job2D &job, const float *factors, float scale, unsigned images, int cu_count, hipStream_t st);
int sg2d_launch_rolling_g3(int n, int terms, const Job2D &job, const float *factors, float scale, unsigned images, int cu_count, hipStream_t st);
int sg2d_launch_rolling_g4(int n, int terms, const Job2D &job, const float *factors, float scale, unsigned images, int cu_count, hipStream_t st);
int sg2d_launch_rolling_g5(int n, int terms, const Job2D &job, const float *factors, float scale, unsigned images, int cu_count, hipStream_t st);
int sg2d_launch_rolling_g6(int n, int terms, const Job2D &job, const float *factors, float scale, unsigned images, int cu_count, hipStream_t st);
inline int sg2d_launch_rolling(int n, int terms, const Job2D &job, const float *factors, float scale, unsigned images, int cu_count,
                               hipStream_t st)
{
    if (sg2d_launch_rolling_g0(n, terms, job, factors, scale, images, cu_count, st) == 0) return 0;
    if (sg2d_launch_rolling_g1(n, terms, job, factors, scale, images, cu_count, st) == 0) return 0;
    if (sg2d_launch_rolling_g2(n, terms, job, factors, scale, images, cu_count, st) == 0) return 0;
    if (sg2d_launch_rolling_g3(n, terms, job, factors, scale, images, cu_count, st) == 0) return 0;
    if (sg2d_launch_rolling_g4(n, terms, job, factors, scale, images, cu_count, st) == 0) return 0;
    if (sg2d_launch_rolling_g5(n, terms, job, factors, scale, images, cu_count, st) == 0) return 0;
    return sg2d_launch_rolling_g6(n, terms, job, factors, scale, images, cu_count, st);
}

// two output frames (job.out, out1) from one walk over the input; both outputs have `terms` (<= 3) terms
int sg2d_launch_rolling2_g0(int n, int terms, const Job2D &job, const float *f0, float s0, const float *f1, float s1, float *out1, unsigned images, int cu_count, hipStream_t st);
int sg2d_launch_rolling2_g1(int n, int terms, const Job2D &job, const float *f0, float s0, const float *f1, float s1, float *out1, unsigned images, int cu_count, hipStream_t st);
int sg2d_launch_rolling2_g2(int n, int terms, const Job2D &job, const float *f0, float s0, const float *f1, float s1, float *out1, unsigned images, int cu_count, hipStream_t st);
int sg2d_launch_rolling2_g3(int n, int terms, const Job2D &job, const float *f0, float s0, const float *f1, float s1, float *out1, unsigned images, int cu_count, hipStream_t st);
int sg2d_launch_rolling2_g4(int n, int terms, const Job2D &job, const float *f0, float s0, const float *f1, float s1, float *out1, unsigned images, int cu_count, hipStream_t st);
int sg2d_launch_rolling2_g5(int n, int terms, const Job2D &job, const float *f0, float s0, const float *f1, float s1, float *out1, unsigned images, int cu_count, hipStream_t st);
int sg2d_launch_rolling2_g6(int n, int terms, const Job2D &job, const float *f0, float s0, const float *f1, float s1, float *out1, unsigned images, int cu_count, hipStream_t st);
inline int sg2d_launch_rolling2(int n, int terms, const Job2D &job, const float *f0, float s0, const float *f1, float s1, float *out1,
                                unsigned images, int cu_count, hipStream_t st)
{
    if (sg2d_launch_rolling2_g0(n, terms, job, f0, s0, f1, s1, out1, images, cu_count, st) == 0) return 0;
    if (sg2d_launch_rolling2_g1(n, terms, job, f0, s0, f1, s1, out1, images, cu_count, st) == 0) return 0;
    if (sg2d_launch_rolling2_g2(n, terms, job, f0, s0, f1, s1, out1, images, cu_count, st) == 0) return 0;
    if (sg2d_launch_rolling2_g3(n, terms, job, f0, s0, f1, s1, out1, images, cu_count, st) == 0) return 0;
    if (sg2d_launch_rolling2_g4(n, terms, job, f0, s0, f1, s1, out1, images, cu_count, st) == 0) return 0;
    if (sg2d_launch_rolling2_g5(n, terms, job, f0, s0, f1, s1, out1, images, cu_count, st) == 0) return 0;
    return sg2d_launch_rolling2_g6(n, terms, job, f0, s0, f1, s1, out1, images, cu_count, st);
}

// sg_2d_hf.hip: ONE term of a kernel with the HORIZONTAL pass first (kernels whose x factor cancels harder than their y factor: deriv_x >= 2,
// deriv_x > deriv_y); job.accumulate: out += result.  Built in three half-window groups (Makefile).  0 = launched, 1 = not covered, -1 = error.
int sg2d_launch_rolling_hf_g0(int n, const Job2D &job, const float *factors, float scale, float sigma, const float *factors2, float sigma2, unsigned images, int cu_count, hipStream_t st);
int sg2d_launch_rolling_hf_g1(int n, const Job2D &job, const float *factors, float scale, float sigma, const float *factors2, float sigma2, unsigned images, int cu_count, hipStream_t st);
int sg2d_launch_rolling_hf_g2(int n, const Job2D &job, const float *factors, float scale, float sigma, const float *factors2, float sigma2, unsigned images, int cu_count, hipStream_t st);
// factors2 / sigma2: a second term in the same launch (NULL: one term); 1 = not covered (the caller launches the terms one by one)
inline int sg2d_launch_rolling_hf(int n, const Job2D &job, const float *factors, float scale, float sigma, const float *factors2, float sigma2, unsigned images, int cu_count,
                                  hipStream_t st)
{
    if (sg2d_launch_rolling_hf_g0(n, job, factors, scale, sigma, factors2, sigma2, images, cu_count, st) == 0) return 0;
    if (sg2d_launch_rolling_hf_g1(n, job, factors, scale, sigma, factors2, sigma2, images, cu_count, st) == 0) return 0;
    return sg2d_launch_rolling_hf_g2(n, job, factors, scale, sigma, factors2, sigma2, images, cu_count, st);
}
// which pass order suits a kernel in fp32 (sg_2d_hf.hip's header, tools/emulate_2d_passes.py, tools/gx_margin_probe.py): the pass that cancels harder -- the
// derivative of higher order, FIRST derivatives included since the end of round 6 -- goes first
inline bool sg2d_x_dominant(int deriv_x, int deriv_y) { return deriv_x >= 1 && deriv_x > deriv_y; }

// sg_2d_dense.hip: the bit-exact dense kernel on packed math, square and rectangular windows with half windows <= DENSE_ROLL_MAX_N.
// 0 = launched, 1 = not covered, -1 = error.  h_w = the kernel on the host, [2ny+1][2nx+1].
constexpr int DENSE_ROLL_MAX_N = 16;
struct DeviceCtx;
int sg2d_launch_dense_rolling(const Job2D &job, const float *h_w, DeviceCtx *ctx, unsigned images, hipStream_t st);

// sg_2d_sep.hip
int sg2d_kernel_double(const Savgol2DConfig *cfg, double *Wd);                           // W in double, [2n+1][2n+1]; 0 on success
int sg2d_factors_from_kernel(const double *Wd, int n, int order, float *factors, int max_terms);   // #terms, 0 = failed
int sg2d_separable_factors(const Savgol2DConfig *cfg, float *factors, int max_terms);   // returns #terms, 0 = not separable here
int sg2d_launch_separable(int n, const Job2D &job, const SepPlan &plan, const float *d_factors, unsigned images, int cu_count,
                          hipStream_t st);

}  // namespace sg
