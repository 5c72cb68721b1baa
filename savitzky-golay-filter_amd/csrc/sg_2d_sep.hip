// sg_2d_sep.hip -- the fast 2-D path: exact low-rank separable passes, fused in one kernel.
//
// The reference's 2-D kernel (src/savgol2d.c:188-265) is a bivariate polynomial of total degree <= order
// sampled on the window, hence a matrix of rank <= order+1:   W[y][x] = sum_t G_t(y) * Q_t(x).
// The host builds the factors in double (orthonormal polynomials G_t in y, Q_t = G_t^T W, terms with a
// vanishing Q_t dropped: r = 2 for the BASELINE config n=7, order 3, d=(0,0)) and the kernel applies
//      out = scale * sum_t  colconv(G_t, rowconv(Q_t, in))
// i.e. 2 r (2n+1) multiply-adds per pixel instead of (2n+1)^2 (60 vs 225 at n=7), which brings the path
// back under the HBM roofline.  Results differ from the dense single-accumulator sum of the reference only
// by fp32 rounding (1e-7 level; tests bound it at 1e-6 normwise against the double-accumulation oracle).
//
// One block = one 64-wide x TH-tall output tile (TH = (64-2N)&~3, so the input tile has <= 64 rows):
//   1. input tile + halo -> LDS (boundary remap applied here: VALID / CONSTANT / REFLECT, reference :428-445)
//   2. per term t:  row pass   (one lane = 16 consecutive outputs of one tile row, sliding window in
//                               registers, v_pk_fma_f32, taps in VGPR pairs)          -> LDS
//                   column pass (one lane = one column x RY consecutive rows, same inner product on a
//                               window gathered down the column) accumulated in registers over the terms
//   3. scaled results -> HBM, 256-B rows per wave-instruction.
// Square windows only (N = nx = ny, compile-time so every tap index is a literal); other shapes run the rolling kernels on
// zero-padded factors, ranks above 4 the dense kernel (sg_2d_dense.hip).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "sg_2d.hpp"
#include "sg_pk.hpp"
#include "sg_runtime.hpp"

namespace sg {

// acc[r] += sum_k w[k] * x[r + k], r < R, k <= 2N, over a window of R + 2N inputs delivered as float4 quads.
// Input stationary, packed: pair j of accumulators takes  w[k] * (x[i], x[i+1])  with i = 2j + k.
template <int N, int R>
struct WinConv {
    static constexpr int NQ = (R + 2 * N + 3) / 4;          // quads in the window
    template <int I, int J = 0>
    static __device__ __forceinline__ void feed(f32x2 (&A)[R / 2], const f32x2 (&W)[N + 1], const f32x2 x)
    {
        if constexpr (J < R / 2) {
            constexpr int k = I - 2 * J;
            if constexpr (k >= 0 && k <= 2 * N) pk_fma_vgpr<(k & 1)>(A[J], W[k >> 1], x);
            feed<I, J + 1>(A, W, x);
        }
    }
    template <int Q, typename LoadQuad>
    static __device__ __forceinline__ void quads(LoadQuad load, f32x2 (&A)[R / 2], const f32x2 (&W)[N + 1], f32x2 prev)
    {
        if constexpr (Q < NQ) {
            const float4 v = load(Q);
            const f32x2 e0 = {v.x, v.y}, e1 = {v.z, v.w};
            if constexpr (Q > 0) feed<4 * Q - 1>(A, W, pk_straddle(prev, e0));
            feed<4 * Q>(A, W, e0);
            feed<4 * Q + 1>(A, W, pk_straddle(e0, e1));
            feed<4 * Q + 2>(A, W, e1);
            quads<Q + 1>(load, A, W, e1);
        }
    }
};

template <int N>
struct Sep {
    static constexpr int TW = 64;
    static constexpr int TH = (64 - 2 * N) & ~3;
    static constexpr int ROWS = TH + 2 * N;                 // <= 64 input rows
    static constexpr int RY = TH / 4;                       // outputs per lane in the column pass
    static constexpr int RYP = (RY + 1) & ~1;               // padded to an even count (pairs)
    static constexpr int PIN = ((64 + 2 * N + 15) & ~15) + 4;   // input tile pitch (floats): = 4 mod 16 -> the row pass's b128 reads are conflict free
    static constexpr int PH = 64 + 4;                       // intermediate pitch: 68 -> conflict-free column reads
};

// NOUT = number of output frames (compile time: the single-output kernel must not pay for the multi-output loop)
// transposed (round 6): the tile is staged TRANSPOSED in LDS (LDS row = image column), so the "row pass" runs down the image columns with
// the G taps and the "column pass" along the image rows with the Q taps -- the VERTICAL pass first, for kernels whose y factor cancels
// harder than their x factor (deriv_y >= 2, deriv_y > deriv_x): the cancelling pass must not be the last arithmetic an output sees
// (sg_2d_hf.hip's header).  An output tile is then TH columns x 64 rows; loads stay coalesced along image rows, the stores are 48-byte
// runs per lane -- the accurate order for method 3 on such kernels, not a fast path (method 2 / auto run them on the rolling kernel).
template <int N, int NOUT>
__global__ __launch_bounds__(256) void sg2d_separable_kernel(const Job2D job, const SepPlan plan, const float *__restrict__ factors,
                                                             unsigned total_tiles, int transposed)
{
    typedef Sep<N> S;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *tin = lds;                                   // [ROWS][PIN]
    float *hbuf = tin + S::ROWS * S::PIN;               // [ROWS][PH]
    float *wl = hbuf + (S::ROWS + 4) * S::PH;           // (4 spare rows, see the column pass) per term: Q_t, G_t (2N+2 floats each)

    const int tid = threadIdx.x;
    constexpr int TCOLS = 64 + 2 * N;

    // persistent blocks; blocks that share an XCD (blockIdx % 8) walk neighbouring tiles (halo reuse in L2)
    const unsigned nblk = gridDim.x;
    const unsigned blk = (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3);
    const unsigned tiles_per_image = (unsigned)(job.tiles_x * job.tiles_y);

    int all_terms = 0;
#pragma unroll
    for (int o = 0; o < NOUT; ++o) all_terms += plan.terms[o];
    for (int i = tid; i < all_terms * 2 * (2 * N + 2); i += 256) wl[i] = factors[i];

    // input tile of tile id t: frame rows y0-N .. y0+TH+N-1, columns x0-N .. x0+63+N, remapped at the frame
    // border (reference savgol2d.c:428-445).  Loaded into registers one tile ahead of the arithmetic.
    // Input tile of tile id t: frame rows y0-N .. y0+TH+N-1, columns x0-N .. x0+63+N, remapped at the frame
    // border (reference savgol2d.c:428-445), loaded into registers one tile ahead of the arithmetic.
    // Wave w takes tile rows w, w+4, ...: the row index (and its remap) is scalar, a lane keeps its two
    // remapped column indices for the whole tile -- no per-element division or remap.
    constexpr int RPW = (S::ROWS + 3) / 4;              // rows per wave
    constexpr int CPW = (TCOLS + 3) / 4;                // transposed: image rows per wave (<= 2 RPW: they share pre0 / pre1)
    static_assert(CPW <= 2 * RPW, "transposed staging re-uses the two prefetch arrays");
    float pre0[RPW], pre1[RPW];
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto prefetch = [&](unsigned t) {
        const unsigned img = t / tiles_per_image, rem = t - img * tiles_per_image;
        const int by = (int)(rem / (unsigned)job.tiles_x), bx = (int)(rem - (unsigned)by * (unsigned)job.tiles_x);
        const float *in = job.in + (long long)img * job.in_pitch;
        if (transposed) {                                // uniform: LDS row = image column x0 - N + lane, LDS column c = image row y0 - N + c
            const int x0 = bx * S::TH, y0 = by * S::TW;
            const int ix = fix_index(x0 + lane - N, job.cols, job.boundary);
#pragma unroll
            for (int j = 0; j < CPW; ++j) {
                const int c = wv + 4 * j;
                if (c < TCOLS && lane < S::ROWS) {
                    const float v = in[(long long)fix_index(y0 + c - N, job.rows, job.boundary) * job.in_stride + ix];
                    if (j < RPW) pre0[j < RPW ? j : 0] = v; else pre1[j >= RPW ? j - RPW : 0] = v;
                }
            }
            return;
        }
        const int x0 = bx * S::TW, y0 = by * S::TH;
        const int ix0 = fix_index(x0 + lane - N, job.cols, job.boundary);
        const int ix1 = fix_index(x0 + lane + 64 - N, job.cols, job.boundary);
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int r = wv + 4 * j;
            if (r < S::ROWS) {                           // uniform
                const float *row = in + (long long)fix_index(y0 + r - N, job.rows, job.boundary) * job.in_stride;
                pre0[j] = row[ix0];
                if (lane + 64 < TCOLS) pre1[j] = row[ix1];
            }
        }
    };

    unsigned tile = blk;
    if (tile < total_tiles) prefetch(tile);

    const int crow = (tid >> 6) * S::RY;                // column pass: this lane's first output row (tile coords)
    const int ccol = tid & 63;

    while (tile < total_tiles) {
        const unsigned img = tile / tiles_per_image, rem = tile - img * tiles_per_image;
        const int by = (int)(rem / (unsigned)job.tiles_x), bx = (int)(rem - (unsigned)by * (unsigned)job.tiles_x);
        const int x0 = bx * (transposed ? S::TH : S::TW), y0 = by * (transposed ? S::TW : S::TH);

        __syncthreads();                                // previous tile's passes are done with tin / hbuf
        if (transposed) {
#pragma unroll
            for (int j = 0; j < CPW; ++j) {
                const int c = wv + 4 * j;
                if (c < TCOLS && lane < S::ROWS) tin[lane * S::PIN + c] = j < RPW ? pre0[j < RPW ? j : 0] : pre1[j >= RPW ? j - RPW : 0];
            }
        } else {
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int r = wv + 4 * j;
            if (r < S::ROWS) {
                tin[r * S::PIN + lane] = pre0[j];
                if (lane + 64 < TCOLS) tin[r * S::PIN + lane + 64] = pre1[j];
            }
        }
        }
        const unsigned next = tile + nblk;
        if (next < total_tiles) prefetch(next);

        int tbase = 0;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {                  // every output re-uses the tile that is already in LDS
        f32x2 acc[S::RYP / 2];
#pragma unroll
        for (int j = 0; j < S::RYP / 2; ++j) acc[j] = f32x2{0.0f, 0.0f};

        for (int t = tbase; t < tbase + plan.terms[o]; ++t) {
            __syncthreads();                            // tin ready / previous term's hbuf consumed
            const float *wt = wl + t * 2 * (2 * N + 2);
            const float *w_first = transposed ? wt + (2 * N + 2) : wt, *w_second = transposed ? wt : wt + (2 * N + 2);     // taps of the LDS-row pass, of the LDS-column pass
            // 2a. row pass: item = (tile row, 16-column segment); lane slides over 16 + 2N inputs
            {
                f32x2 Wq[N + 1];
#pragma unroll
                for (int p = 0; p < N + 1; ++p) Wq[p] = f32x2{w_first[2 * p], w_first[2 * p + 1]};
                if (tid < S::ROWS * 4) {
                    const int r = tid >> 2, seg = tid & 3;
                    const float *src = tin + r * S::PIN + seg * 16;
                    f32x2 A[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) A[j] = f32x2{0.0f, 0.0f};
                    // derivative kernels (plan.centre): the taps meet centred samples, s - c with c = the sample in the middle of the segment's 16 + 2N
                    // inputs, and c times what the factor sums to in the reference's dense table is added back -- an offset, a ramp or a slow swing
                    // under the signal then never meets the cancelling taps (and their fp32 rounding) at full size (sg_2d_hf.hip, EXPERIMENTS R6.8)
                    const float c = plan.centre ? src[N + 8] : 0.0f;
                    WinConv<N, 16>::template quads<0>([&](int q) { const float4 v = *reinterpret_cast<const float4 *>(src + 4 * q);
                                                                    return make_float4(v.x - c, v.y - c, v.z - c, v.w - c); }, A, Wq, f32x2{0.0f, 0.0f});
                    if (plan.centre) {
                        const float back = c * plan.first_sum[t - tbase];
#pragma unroll
                        for (int j = 0; j < 8; ++j) A[j] = A[j] + f32x2{back, back};
                    }
                    float *dst = hbuf + r * S::PH + seg * 16;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<float4 *>(dst + 4 * j) = make_float4(A[2 * j].x, A[2 * j].y, A[2 * j + 1].x, A[2 * j + 1].y);
                }
            }
            __syncthreads();
            // 2b. column pass: lane = column ccol, rows crow .. crow+RY-1; window gathered down the column
            {
                f32x2 Wg[N + 1];
#pragma unroll
                for (int p = 0; p < N + 1; ++p) Wg[p] = f32x2{w_second[2 * p], w_second[2 * p + 1]};
                const float *col = hbuf + crow * S::PH + ccol;
                WinConv<N, S::RYP>::template quads<0>(
                    [&](int q) {
                        const int r0 = 4 * q;
                        if constexpr (S::RYP == S::RY) {
                            // every needed row is inside the tile; the last quad may run up to 3 rows past it, into the
                            // spare rows kept behind hbuf (their values only meet taps that do not exist)
                            return make_float4(col[r0 * S::PH], col[(r0 + 1) * S::PH], col[(r0 + 2) * S::PH], col[(r0 + 3) * S::PH]);
                        } else {
                            // rows beyond the tile (touched only by the padded accumulator of an odd RY or by the last,
                            // partly unused quad) read row ROWS-1 instead
                            auto rd = [&](int r) { return col[(crow + r < S::ROWS ? r : S::ROWS - 1 - crow) * S::PH]; };
                            return make_float4(rd(r0), rd(r0 + 1), rd(r0 + 2), rd(r0 + 3));
                        }
                    },
                    acc, Wg, f32x2{0.0f, 0.0f});
            }
        }

        // 3. store
        float *out = plan.out[o] + (long long)img * job.out_pitch;
        const bool valid = job.boundary == SAVGOL2D_BOUNDARY_VALID;
        const int xlo = valid ? N : 0, xhi = valid ? job.cols - N : job.cols;
        const int ylo = valid ? N : 0, yhi = valid ? job.rows - N : job.rows;
#pragma unroll
        for (int j = 0; j < S::RY; ++j) {
            const int ox = transposed ? x0 + crow + j : x0 + ccol, oy = transposed ? y0 + ccol : y0 + crow + j;
            const float v = (j & 1) ? acc[j >> 1].y : acc[j >> 1].x;
            if (ox >= xlo && ox < xhi && oy >= ylo && oy < yhi) out[(long long)oy * job.out_stride + ox] = v * plan.scale[o];
        }
        tbase += plan.terms[o];
        }
        tile = next;
    }
}

template <int N>
static void launch_sep(const Job2D &job, const SepPlan &plan, const float *d_factors, unsigned images, int cu_count, hipStream_t st)
{
    typedef Sep<N> S;
    Job2D j = job;
    const int transposed = plan.transposed;
    j.tiles_x = (job.cols + (transposed ? S::TH : S::TW) - 1) / (transposed ? S::TH : S::TW);
    j.tiles_y = (job.rows + (transposed ? S::TW : S::TH) - 1) / (transposed ? S::TW : S::TH);
    const size_t lds = sizeof(float) * (S::ROWS * S::PIN + (S::ROWS + 4) * S::PH + SEP_MAX_OUTPUTS * SEP_MAX_TERMS * 2 * (2 * N + 2));
    const unsigned long long total = (unsigned long long)images * j.tiles_x * j.tiles_y;      // caller keeps this < 2^32
    const unsigned per_cu = (unsigned)(160 * 1024 / lds) < 4u ? (unsigned)(160 * 1024 / lds) : 4u;
    unsigned grid = (unsigned)cu_count * (per_cu ? per_cu : 1u);
    if (grid > total) grid = (unsigned)total;
    grid = (grid + 7u) & ~7u;
    // one output frame per launch since round 6 (the callers loop: the multi-output instantiations were two thirds of this object and only a fallback's fallback)
    hipLaunchKernelGGL((sg2d_separable_kernel<N, 1>), dim3(grid), dim3(256), lds, st, j, plan, d_factors, (unsigned)total, transposed);
}

template <int N>
static int dispatch_sep(int n, const Job2D &job, const SepPlan &plan, const float *d_factors, unsigned images, int cu_count, hipStream_t st)
{
    if (n == N) { launch_sep<N>(job, plan, d_factors, images, cu_count, st); return 1; }
    if constexpr (N < SAVGOL2D_MAX_HALF_WINDOW) return dispatch_sep<N + 1>(n, job, plan, d_factors, images, cu_count, st);
    else return 0;
}

int sg2d_launch_separable(int n, const Job2D &job, const SepPlan &plan, const float *d_factors, unsigned images, int cu_count,
                          hipStream_t st)
{
    if (plan.outputs != 1) return -1;
    return dispatch_sep<1>(n, job, plan, d_factors, images, cu_count, st) ? 0 : -1;
}

// ---- host: exact low-rank factors of the least-squares kernel, in double ----
// factors layout per term: Q_t[0..2N] then a pad, G_t[0..2N] then a pad  (2 * (2N+2) floats)
int sg2d_kernel_double(const Savgol2DConfig *cfg, double *Wd)
{
    const int n = cfg->half_window_x, order = cfg->poly_order, ws = 2 * n + 1;
    float wf[SAVGOL2D_MAX_WINDOW_AREA];
    double coef[SAVGOL2D_MAX_TERMS];
    if (cfg->half_window_x != cfg->half_window_y || sg2d_weights_fill(cfg, wf, coef) != 0) return -1;
    // W in double from the polynomial coefficients (already scaled by dx! dy!)
    for (int y = -n; y <= n; ++y)
        for (int x = -n; x <= n; ++x) {
            double s = 0.0;
            for (int px = 0; px <= order; ++px)
                for (int py = 0; px + py <= order; ++py) s += coef[sg2d_term(px, py)] * std::pow((double)x, px) * std::pow((double)y, py);
            Wd[(y + n) * ws + (x + n)] = s;
        }
    return 0;
}

int sg2d_factors_from_kernel(const double *Wd, int n, int order, float *factors, int max_terms)
{
    const int ws = 2 * n + 1;
    double wmax = 0.0;
    for (int i = 0; i < ws * ws; ++i) if (std::fabs(Wd[i]) > wmax) wmax = std::fabs(Wd[i]);
    // orthonormal polynomials in y on the window (modified Gram-Schmidt on 1, y, y^2, ...)
    const int nb = (order + 1 < ws) ? order + 1 : ws;
    double G[7][33];
    int terms = 0;
    for (int j = 0; j < nb; ++j) {
        double v[33];
        for (int y = 0; y < ws; ++y) v[y] = std::pow((double)(y - n) / (double)n, j);
        for (int pass = 0; pass < 2; ++pass)
            for (int k = 0; k < j; ++k) {
                double d = 0.0;
                for (int y = 0; y < ws; ++y) d += G[k][y] * v[y];
                for (int y = 0; y < ws; ++y) v[y] -= d * G[k][y];
            }
        double nrm = 0.0;
        for (int y = 0; y < ws; ++y) nrm += v[y] * v[y];
        nrm = std::sqrt(nrm);
        for (int y = 0; y < ws; ++y) G[j][y] = v[y] / nrm;
    }
    for (int j = 0; j < nb; ++j) {
        double Q[33], qmax = 0.0;
        for (int x = 0; x < ws; ++x) {
            double s = 0.0;
            for (int y = 0; y < ws; ++y) s += G[j][y] * Wd[y * ws + x];
            Q[x] = s;
            if (std::fabs(s) > qmax) qmax = std::fabs(s);
        }
        if (qmax <= 1e-13 * wmax) continue;                      // this y-polynomial does not occur in W
        if (terms >= max_terms) return 0;
        float *f = factors + (size_t)terms * 2 * (ws + 1);
        memset(f, 0, sizeof(float) * 2 * (ws + 1));
        for (int x = 0; x < ws; ++x) f[x] = (float)Q[x];
        for (int y = 0; y < ws; ++y) f[(ws + 1) + y] = (float)G[j][y];
        ++terms;
    }
    return terms;
}

// Rectangular windows (nx != ny; reference savgol2d.h:82-90, test_savgol2d.c:508-543): the same factorisation on the
// (2ny+1) x (2nx+1) kernel, G_t orthonormal polynomials on the y window, Q_t = G_t^T W on the x window, each embedded in the
// middle of a vector of 2N+1 entries, N = max(nx, ny), zeros either side -- the square kernels of half window N then apply it
// unchanged (zero taps add nothing; the stored range of VALID comes from the job's own nx, ny).  One caveat, stated in
// savgol_hip.h: a zero tap times a non-finite sample is NaN, so NaN / Inf spread over the padded window in this method.
static int sg2d_rect_factors(const Savgol2DConfig *cfg, float *factors, int max_terms)
{
    const int nx = cfg->half_window_x, ny = cfg->half_window_y, order = cfg->poly_order;
    const int ww = 2 * nx + 1, wh = 2 * ny + 1, N = nx > ny ? nx : ny, ws = 2 * N + 1;
    float wf[SAVGOL2D_MAX_WINDOW_AREA];
    double coef[SAVGOL2D_MAX_TERMS];
    if (sg2d_weights_fill(cfg, wf, coef) != 0) return 0;
    static thread_local double Wd[33 * 33];
    double wmax = 0.0;
    for (int y = -ny; y <= ny; ++y)
        for (int x = -nx; x <= nx; ++x) {
            double s = 0.0;
            for (int px = 0; px <= order; ++px)
                for (int py = 0; px + py <= order; ++py) s += coef[sg2d_term(px, py)] * std::pow((double)x, px) * std::pow((double)y, py);
            Wd[(y + ny) * ww + (x + nx)] = s;
            if (std::fabs(s) > wmax) wmax = std::fabs(s);
        }
    const int nb = (order + 1 < wh) ? order + 1 : wh;
    double G[7][33];
    int terms = 0;
    for (int j = 0; j < nb; ++j) {
        double v[33];
        for (int y = 0; y < wh; ++y) v[y] = std::pow((double)(y - ny) / (double)ny, j);
        for (int pass = 0; pass < 2; ++pass)
            for (int k = 0; k < j; ++k) {
                double d = 0.0;
                for (int y = 0; y < wh; ++y) d += G[k][y] * v[y];
                for (int y = 0; y < wh; ++y) v[y] -= d * G[k][y];
            }
        double nrm = 0.0;
        for (int y = 0; y < wh; ++y) nrm += v[y] * v[y];
        nrm = std::sqrt(nrm);
        for (int y = 0; y < wh; ++y) G[j][y] = v[y] / nrm;
    }
    for (int j = 0; j < nb; ++j) {
        double Q[33], qmax = 0.0;
        for (int x = 0; x < ww; ++x) {
            double s = 0.0;
            for (int y = 0; y < wh; ++y) s += G[j][y] * Wd[y * ww + x];
            Q[x] = s;
            if (std::fabs(s) > qmax) qmax = std::fabs(s);
        }
        if (qmax <= 1e-13 * wmax) continue;
        if (terms >= max_terms) return 0;
        float *f = factors + (size_t)terms * 2 * (ws + 1);
        memset(f, 0, sizeof(float) * 2 * (ws + 1));
        for (int x = 0; x < ww; ++x) f[(N - nx) + x] = (float)Q[x];
        for (int y = 0; y < wh; ++y) f[(ws + 1) + (N - ny) + y] = (float)G[j][y];
        ++terms;
    }
    return terms;
}

int sg2d_separable_factors(const Savgol2DConfig *cfg, float *factors, int max_terms)
{
    if (cfg->half_window_x != cfg->half_window_y) return sg2d_rect_factors(cfg, factors, max_terms);
    static thread_local double Wd[33 * 33];
    if (sg2d_kernel_double(cfg, Wd) != 0) return 0;
    return sg2d_factors_from_kernel(Wd, cfg->half_window_x, cfg->poly_order, factors, max_terms);
}

}  // namespace sg
