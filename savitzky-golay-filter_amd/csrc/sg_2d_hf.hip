// sg_2d_hf.hip -- the rolling-window 2-D kernel with the HORIZONTAL pass first (round 6).
//
// The low-rank form  W(x,y) = sum_t G_t(y) Q_t(x)  (reference kernel: src/savgol2d.c:188-265) can be applied in either order, and in fp32
// the order matters for kernels whose x factor cancels harder than their y factor (deriv_x >= 2 and deriv_x > deriv_y: d^2/dx^2, the
// Hessian's xx frame, ...).  sg_2d_roll.hip runs the vertical pass first: there the x-derivative taps -- which sum to zero -- are the LAST
// arithmetic an output sees, and both the rounding of the smoothed intermediate row and that pass's own rounding reach the output at
// full size: 1.8-3.3e-6 of the output (normwise) where the reference's dense sum, whose inner loop runs along x, is at 0.4-0.9e-6.
// With the cancelling pass FIRST its rounding errors are independent from row to row and the smoothing pass down the column averages them:
// 0.5-0.75e-6 on every (n, order, dx > dy) of tools/emulate_2d_passes.py -- inside max(1e-6, 1.1 x the reference's own error) with no
// wider constant (the y-dominant kernels already had the good order).
//
// Structure: the strip walk of sg_2d_roll.hip with the roles of the two stores swapped.  A wave owns a 256-column strip and walks down a band
// of rows; every input row goes through a wave-private LDS row ONCE, as it arrives (each lane writes its 4 columns, reads back the 4 + 2N
// columns of its window), the lane computes  h = Q (*)x row  for its 4 columns, and the REGISTER RING holds the last 2N+1 rows of h.  The
// vertical pass  out = G (*)y h  then reads the ring with literal slots and stores straight from registers.  One or TWO terms per launch (two: half windows
// <= 8, x factors of the same parity -- the x-derivative frame of an order-3 gradient is x and x^3: both rings are fed from one trip of the row through
// LDS and one set of folded pairs); further terms run one launch each, accumulating (out += ...: 8 + 12 B per pixel and term instead of 8).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "sg_2d.hpp"
#include "sg_pk.hpp"
#include "sg_runtime.hpp"

namespace sg {

template <int N>
struct Hf {
    static constexpr int HL = N <= 4 ? 2 : (N <= 6 ? 4 : (N + 3) / 4);       // halo lanes per side: the strip geometry of sg_2d_roll.hip
    static constexpr int OUTL = 64 - 2 * HL, SW = 4 * OUTL, NQ = 2 * HL + 1, D = 4 * HL - N;
    static constexpr int P = N >= 13 ? 1 : ((N & 1) ? 5 : 3);                 // raw rows in flight ahead of the conversion
    static constexpr int U = 2 * N + 1 + P;                                   // ring slots = unroll factor of the row loop
    static constexpr int PR = N >= 13 ? 2 : 4, LEAD = PR - 1;                 // accumulating pass: ring of prefetched output rows
    static constexpr int BUFW = 256 + 8 * HL;                                 // LDS floats per wave: one row + pads
    static constexpr int NP = N / 2 + 1;
    static constexpr int WPB = 4;
    static_assert(U % PR == 0 && P >= LEAD, "prefetched output rows: slot = row % PR must carry over from one group of U rows to the next");
};

template <int N, int NT>
struct HfTaps {
    f32x2 g[NT][Hf<N>::NP], q[NT][Hf<N>::NP];      // taps 0..N (the mirrored half follows from the parity); q already times the output scale
    f32x2 sy[NT], sx;                               // +1 / -1 (the x parity is common to the terms of one launch)
    f32x2 sig[NT];                                  // what each term's x factor must sum to (both halves equal): see convert()
};
constexpr int HF_TWO_TERMS_MAX_N = 8;               // two rings of h rows: 2 x 4 (2N + 1 + P) registers

// MODE 1: the strip's 256 input columns are inside the frame, all SW output columns are stored, rows 16-byte aligned; 0: remapped scalar loads, masked stores
template <int N, int MODE, bool ACC, int NT>
__device__ __forceinline__ void hf_item(const Job2D &job, const HfTaps<N, NT> &taps, float *mine, const float *in, float *out, int xload, int yb, int nout,
                                        int lane, int xlo, int xhi, int ylo, int yhi)
{
    typedef Hf<N> R;
    constexpr bool VEC = MODE != 0;
    const int c0 = xload + 4 * lane;
    int ix0 = 0, ix1 = 0, ix2 = 0, ix3 = 0;
    if constexpr (!VEC) {
        ix0 = fix_index(c0, job.cols, job.boundary); ix1 = fix_index(c0 + 1, job.cols, job.boundary);
        ix2 = fix_index(c0 + 2, job.cols, job.boundary); ix3 = fix_index(c0 + 3, job.cols, job.boundary);
    }
    const bool reflect = job.boundary == SAVGOL2D_BOUNDARY_REFLECT;
    auto load_row = [&](int r) -> f32x4 {                    // band row r = frame row yb - N + r, remapped at the frame border; rows past the band: clamped re-reads nobody uses
        const float *row = in + (long long)fix_row(yb - N + r, job.rows, reflect) * job.in_stride;
        if constexpr (VEC) return *reinterpret_cast<const f32x4 *>(row + c0);
        else return f32x4{row[ix0], row[ix1], row[ix2], row[ix3]};
    };
    const bool out_lane = lane >= R::HL && lane < 64 - R::HL;
    const int yend = yb + nout;
    float *const wr = mine + 4 * R::HL + 4 * lane;
    const float *const rd = mine + 4 * lane;

    f32x4 win[R::U];          // slot r % U: row r, raw until its conversion, then h (of term 0)
    f32x4 win2[NT == 2 ? R::U : 1];      // h of term 1

    // the horizontal pass of one row: through the LDS row and back, folded taps (the pass of sg_2d_roll.hip's hterm, general form)
    auto convert = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        wave_lds_sync();                                     // the previous row's window reads are ordered before this write
        *reinterpret_cast<f32x4 *>(wr) = win[s];
        wave_lds_sync();
        // CENTRED (round 6, R6.8): sum_k q_k s_k = sum_k q_k (s_k - c) + c sum_k q_k for ANY c.  With c = a sample of the lane's own window the
        // differences are as small as the data's local variation, so an offset or a ramp under the signal no longer meets the cancelling taps at full
        // size; and the last term takes sum_k q_k not from the rounded taps but from the REFERENCE's dense table (sig, fitted on the host: what this
        // term must sum to for the rows of  sum_t G_t(y) sig_t  to equal the row sums of W) -- the systematic part of the taps' rounding.  Emulated
        // (tools/emulate_2d_passes.py --centre): 1.2-2.4e-7 of the output whatever the offset, where the plain pass (and the reference's own loop) grow with it.
        const f32x2 cc = f32x2{win[s].x, win[s].x};
        f32x2 e[2 * R::NQ + 1];
#pragma unroll
        for (int q = 0; q < R::NQ; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(rd + 4 * q);
            e[2 * q] = f32x2{v.x, v.y} - cc;
            e[2 * q + 1] = f32x2{v.z, v.w} - cc;
        }
        e[2 * R::NQ] = f32x2{0.0f, 0.0f};
        f32x2 pr[2 * N + 3];                                 // pr[j] = window floats (D+j, D+j+1)
#pragma unroll
        for (int j = 0; j < 2 * N + 3; ++j) {
            const int idx = R::D + j;
            pr[j] = (idx & 1) ? (((idx >> 1) & 1) ? pk_straddle(e[idx >> 1], e[(idx >> 1) + 1]) : pk_middle(e[idx >> 1], e[(idx >> 1) + 1])) : e[idx >> 1];
        }
        f32x2 f[2][N + 1], r[2], r2[2];
        auto fold = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k < N) {
                f[0][k] = pk_fold(taps.sx, pr[2 * N - k], pr[k]);
                f[1][k] = pk_fold(taps.sx, pr[2 + 2 * N - k], pr[2 + k]);
            } else {
                f[0][N] = pr[N];
                f[1][N] = pr[2 + N];
            }
        };
        fold(std::integral_constant<int, 0>{});
        static_for(std::make_integer_sequence<int, N + 1>{}, [&](auto kc) -> bool {
            constexpr int k = decltype(kc)::value;
            if constexpr (k < N) fold(std::integral_constant<int, k + 1>{});
            if constexpr (k == 0) { r[0] = pk_mul_sgpr<0>(taps.q[0][0], f[0][0]); r[1] = pk_mul_sgpr<0>(taps.q[0][0], f[1][0]); }
            else { pk_fma_sgpr<(k & 1)>(r[0], taps.q[0][k >> 1], f[0][k]); pk_fma_sgpr<(k & 1)>(r[1], taps.q[0][k >> 1], f[1][k]); }
            if constexpr (NT == 2) {                         // the second term's taps on the same folded pairs
                if constexpr (k == 0) { r2[0] = pk_mul_sgpr<0>(taps.q[1][0], f[0][0]); r2[1] = pk_mul_sgpr<0>(taps.q[1][0], f[1][0]); }
                else { pk_fma_sgpr<(k & 1)>(r2[0], taps.q[1][k >> 1], f[0][k]); pk_fma_sgpr<(k & 1)>(r2[1], taps.q[1][k >> 1], f[1][k]); }
            }
            return true;
        });
        r[0] = __builtin_elementwise_fma(taps.sig[0], cc, r[0]);
        r[1] = __builtin_elementwise_fma(taps.sig[0], cc, r[1]);
        win[s] = f32x4{r[0].x, r[0].y, r[1].x, r[1].y};
        if constexpr (NT == 2) {
            r2[0] = __builtin_elementwise_fma(taps.sig[1], cc, r2[0]);
            r2[1] = __builtin_elementwise_fma(taps.sig[1], cc, r2[1]);
            win2[s] = f32x4{r2[0].x, r2[0].y, r2[1].x, r2[1].y};
        }
    };
    // the vertical pass of the output row whose first h row sits in slot u0: sum over the launch's terms of G_t (*) h_t
    auto vertical = [&](auto u0c) -> f32x4 {
        constexpr int u0 = decltype(u0c)::value;
        f32x2 v[2];
        static_for<NT>([&](auto tc) -> bool {
            constexpr int t = decltype(tc)::value;
            f32x2 f[2][N + 1];
            auto fold = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                f32x4 a, b;
                if constexpr (t == 0) { a = win[(u0 + k) % R::U]; b = win[(u0 + 2 * N - k) % R::U]; }
                else                  { a = win2[(u0 + k) % R::U]; b = win2[(u0 + 2 * N - k) % R::U]; }
                if constexpr (k < N) {
                    f[0][k] = pk_fold(taps.sy[t], f32x2{b.x, b.y}, f32x2{a.x, a.y});
                    f[1][k] = pk_fold(taps.sy[t], f32x2{b.z, b.w}, f32x2{a.z, a.w});
                } else {
                    f[0][N] = f32x2{a.x, a.y};
                    f[1][N] = f32x2{a.z, a.w};
                }
            };
            fold(std::integral_constant<int, 0>{});
            static_for(std::make_integer_sequence<int, N + 1>{}, [&](auto kc) -> bool {
                constexpr int k = decltype(kc)::value;
                if constexpr (k < N) fold(std::integral_constant<int, k + 1>{});
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if constexpr (k == 0 && t == 0) v[c] = pk_mul_sgpr<0>(taps.g[t][0], f[c][0]);
                    else pk_fma_sgpr<(k & 1)>(v[c], taps.g[t][k >> 1], f[c][k]);
                }
                return true;
            });
            return true;
        });
        return f32x4{v[0].x, v[0].y, v[1].x, v[1].y};
    };
    f32x4 prevq[R::PR];
    auto load_prev = [&](int yo) -> f32x4 {                  // ACC: what the output frame holds in this lane's columns of frame row yo
        const float *orow = out + (long long)(yo < job.rows ? yo : job.rows - 1) * job.out_stride;
        if constexpr (VEC) return *reinterpret_cast<const f32x4 *>(orow + c0);
        else {
            const bool row_ok = yo >= ylo && yo < yhi;
            auto at = [&](int c) -> float { return (row_ok && out_lane && c >= xlo && c < xhi) ? orow[c] : 0.0f; };
            return f32x4{at(c0), at(c0 + 1), at(c0 + 2), at(c0 + 3)};
        }
    };
    auto store_row = [&](const f32x4 r, int yo) {
        if (yo >= ylo && yo < yhi && yo < yend) {            // uniform
            float *orow = out + (long long)yo * job.out_stride;
            if constexpr (VEC) {
                if (out_lane) __builtin_nontemporal_store(__builtin_bit_cast(u32x4, r), reinterpret_cast<u32x4 *>(orow + c0));
            } else if (out_lane) {
                if (c0 >= xlo && c0 < xhi) orow[c0] = r.x;
                if (c0 + 1 >= xlo && c0 + 1 < xhi) orow[c0 + 1] = r.y;
                if (c0 + 2 >= xlo && c0 + 2 < xhi) orow[c0 + 2] = r.z;
                if (c0 + 3 >= xlo && c0 + 3 < xhi) orow[c0 + 3] = r.w;
            }
        }
    };

#pragma unroll
    for (int r = 0; r < R::U; ++r) win[r] = load_row(r);
    if constexpr (ACC) {
#pragma unroll
        for (int j = 0; j < R::LEAD; ++j) prevq[j] = load_prev(yb + j);
    }
    static_for<2 * N>([&](auto rc) -> bool { convert(rc); return true; });          // rows 0 .. 2N-1; row 2N is converted by the first iteration
    // iteration m: convert row m + 2N, then output row m = G (*) h rows m .. m + 2N.  Slot of row r = r % U; base is a multiple of U.
    for (int base = 0; base < nout; base += R::U) {
        const bool go = static_for(std::make_integer_sequence<int, R::U>{}, [&](auto uuc) -> bool {
            constexpr int uu = decltype(uuc)::value;
            const int m = base + uu;
            if (m >= nout) return false;                     // uniform
            if (m > 0) win[(uu + R::U - 1) % R::U] = load_row(m + R::U - 1);     // the slot row m-1 left (m = 0: row U-1 is already on its way)
            if constexpr (ACC) prevq[(uu + R::LEAD) % R::PR] = load_prev(yb + m + R::LEAD);
            convert(std::integral_constant<int, (uu + 2 * N) % R::U>{});
            f32x4 r = vertical(uuc);
            if constexpr (ACC) r = r + prevq[uu % R::PR];
            store_row(r, yb + m);
            return true;
        });
        if (!go) break;
    }
}

constexpr int hf_min_waves(int n) { return n <= 10 ? 2 : 1; }

template <int N, bool ACC, int NT>
__global__ __launch_bounds__(64 * Hf<N>::WPB, NT == 2 ? 1 : hf_min_waves(N)) void sg2d_rolling_hf_kernel(const Job2D job, const HfTaps<N, NT> taps, unsigned strips, unsigned bands,
                                                                                              int band_rows, unsigned total_items, int aligned)
{
    typedef Hf<N> R;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *mine = lds + wv * R::BUFW;
    // blocks that share an XCD (blockIdx % 8, observed placement; speed only) take neighbouring items
    const unsigned nblk = gridDim.x;
    const unsigned blk = (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3);
    const unsigned item = blk * R::WPB + (unsigned)wv;
    if (item >= total_items) return;
    const bool valid = job.boundary == SAVGOL2D_BOUNDARY_VALID;
    const int xlo = valid ? job.nx : 0, xhi = valid ? job.cols - job.nx : job.cols;
    const int ylo = valid ? job.ny : 0, yhi = valid ? job.rows - job.ny : job.rows;
    const unsigned strip = item % strips, ib = item / strips;
    const unsigned band = ib % bands, img = ib / bands;
    const int yb = (int)band * band_rows;
    const int nout = job.rows - yb < band_rows ? job.rows - yb : band_rows;
    const float *in = job.in + (long long)img * job.in_pitch;
    float *out = job.out + (long long)img * job.out_pitch;
    const int sx = (int)strip * R::SW;
    if (aligned == 3 && sx - 4 * R::HL >= 0 && sx - 4 * R::HL + 256 <= job.cols && sx >= xlo && sx + R::SW <= xhi)
        hf_item<N, 1, ACC, NT>(job, taps, mine, in, out, sx - 4 * R::HL, yb, nout, lane, xlo, xhi, ylo, yhi);
    else
        hf_item<N, 0, ACC, NT>(job, taps, mine, in, out, sx - 4 * R::HL, yb, nout, lane, xlo, xhi, ylo, yhi);
}

// ---- host ----
// term t of the launch's taps; false: no definite parity, or (t = 1) an x parity other than term 0's
template <int N, int NT>
static bool hf_fill_taps(HfTaps<N, NT> &taps, int t, const float *factors, float scale, float sigma)
{
    taps.sig[t] = f32x2{sigma, sigma};
    const float *q = factors, *g = q + (2 * N + 2);
    float sy, sx;
    if (!vector_parity(g, N, &sy) || !vector_parity(q, N, &sx)) return false;
    if (t > 0 && sx != taps.sx.x) return false;
    for (int k = 0; k <= N; ++k) {
        const float gk = (k == N && sy < 0.0f) ? 0.0f : g[k];
        const float qk = (k == N && sx < 0.0f) ? 0.0f : (float)((double)q[k] * (double)scale);
        if (k & 1) { taps.g[t][k >> 1].y = gk; taps.q[t][k >> 1].y = qk; }
        else       { taps.g[t][k >> 1].x = gk; taps.q[t][k >> 1].x = qk; }
    }
    taps.sy[t] = f32x2{sy, sy};
    taps.sx = f32x2{sx, sx};
    return true;
}

template <int N, bool ACC, int NT>
static int hf_launch(const Job2D &job, const HfTaps<N, NT> &taps, unsigned images, int cu_count, hipStream_t st)
{
    typedef Hf<N> R;
    int aligned = 0;
    if (job.in_stride % 4 == 0 && job.in_pitch % 4 == 0 && (reinterpret_cast<uintptr_t>(job.in) & 15u) == 0) aligned |= 1;
    if (job.out_stride % 4 == 0 && job.out_pitch % 4 == 0 && (reinterpret_cast<uintptr_t>(job.out) & 15u) == 0) aligned |= 2;
    const unsigned strips = (unsigned)((job.cols + R::SW - 1) / R::SW);
    const size_t lds = sizeof(float) * R::WPB * R::BUFW;
    static int per_cu = 0;
    if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sg2d_rolling_hf_kernel<N, ACC, NT>, 64 * R::WPB, lds) != hipSuccess || nb < 1) nb = 1;
        per_cu = nb > 4 ? 4 : nb;
    }
    const unsigned nwaves = (unsigned)cu_count * (unsigned)per_cu * R::WPB;
    unsigned long long per_image;
    unsigned bands;
    int band_rows;
    auto geometry = [&](unsigned long long imgs) {
        bands = choose_bands(job.rows, imgs * strips, nwaves, N, 0.6);       // warm-up rows are loaded AND converted
        band_rows = (int)((job.rows + (int)bands - 1) / (int)bands);
        bands = (unsigned)((job.rows + band_rows - 1) / band_rows);
        per_image = (unsigned long long)strips * bands;
    };
    geometry(images);
    const unsigned long long max_items = ((1ull << 32) - 4096) / 64;
    unsigned long long img_step = per_image ? max_items / per_image : images;
    if (img_step == 0) { sg_set_error("2-D frame too large for one launch (%llu items)", per_image); return -1; }
    if (img_step < images) geometry(img_step); else img_step = images;
    for (unsigned long long i0 = 0; i0 < images; i0 += img_step) {
        const unsigned long long ni = images - i0 < img_step ? images - i0 : img_step;
        const unsigned long long total = ni * per_image;
        unsigned grid = (unsigned)((total + R::WPB - 1) / R::WPB);
        grid = (grid + 7u) & ~7u;
        Job2D part = job;
        part.in = job.in + (long long)i0 * job.in_pitch;
        part.out = job.out + (long long)i0 * job.out_pitch;
        hipLaunchKernelGGL((sg2d_rolling_hf_kernel<N, ACC, NT>), dim3(grid), dim3(64 * R::WPB), lds, st, part, taps, strips, bands, band_rows, (unsigned)total, aligned);
    }
    return 0;
}

template <int N>
static int hf_dispatch(int n, const Job2D &job, const float *factors, float scale, float sigma, const float *factors2, float sigma2, unsigned images, int cu_count,
                       hipStream_t st)
{
    if (n == N) {
        if (factors2) {
            if constexpr (N <= HF_TWO_TERMS_MAX_N) {
                if (job.accumulate) return 1;
                HfTaps<N, 2> taps;
                memset(&taps, 0, sizeof(taps));
                if (!hf_fill_taps<N, 2>(taps, 0, factors, scale, sigma) || !hf_fill_taps<N, 2>(taps, 1, factors2, scale, sigma2)) return 1;
                return hf_launch<N, false, 2>(job, taps, images, cu_count, st);
            } else return 1;
        }
        HfTaps<N, 1> taps;
        memset(&taps, 0, sizeof(taps));
        if (!hf_fill_taps<N, 1>(taps, 0, factors, scale, sigma)) return 1;
        return job.accumulate ? hf_launch<N, true, 1>(job, taps, images, cu_count, st) : hf_launch<N, false, 1>(job, taps, images, cu_count, st);
    }
    if constexpr (N < SG_HF_MAX_N) return hf_dispatch<N + 1>(n, job, factors, scale, sigma, factors2, sigma2, images, cu_count, st);
    else return 1;
}

// ONE term (factors: Q[0..2N], pad, G[0..2N], pad) of a kernel, horizontal pass first; job.accumulate: out += result.  sigma: what this term's scaled x
// factor sums to in the reference's dense table (sg_2d.hip: hf_term_sums).  factors2 / sigma2 (may be NULL): a SECOND term in the same launch (half
// windows <= 8, both x factors of one parity, not accumulating).
// 0 = launched, 1 = not covered by this object (half window outside SG_HF_MIN_N..SG_HF_MAX_N, no definite parity, no two-term form), -1 = error.
int SG_HF_FN(int n, const Job2D &job, const float *factors, float scale, float sigma, const float *factors2, float sigma2, unsigned images, int cu_count,
             hipStream_t st)
{
    if (n < SG_HF_MIN_N || n > SG_HF_MAX_N) return 1;
    return hf_dispatch<SG_HF_MIN_N>(n, job, factors, scale, sigma, factors2, sigma2, images, cu_count, st);
}

}  // namespace sg
