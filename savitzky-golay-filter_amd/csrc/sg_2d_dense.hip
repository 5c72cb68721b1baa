// sg_2d_dense.hip -- the bit-exact 2-D path (every window 3 x 3 .. 33 x 33, square or rectangular), on packed math.
//
// Reference arithmetic (src/savgol2d.c:374-393, :417-453): one fp32 accumulator per output pixel, the window walked
// row-major (wy outer, wx inner), `sum += W[wy][wx] * in[..]` with multiply and add rounded separately, then * scale.
// That order only constrains each output on its own: its input rows must arrive in ascending order and, inside a row,
// the taps in ascending wx.  So the kernel is input-row stationary: one WAVE owns a 256-column strip (a lane owns 4
// adjacent columns = two packed pairs) and walks down a band of rows; when input row r arrives it is fed, with
// W[wy][:], into the accumulators of all 2N+1 output rows r-wy that see it.  2N+1 accumulator pairs per column pair
// live in registers and shift down by one when a row completes; the input row crosses lanes once, through a
// wave-private LDS row; the 2N+1 taps of one W row at a time come in through scalar loads.  v_pk_mul_f32 +
// v_pk_add_f32: two pixels per instruction, the reference's two roundings per tap -> bit-identical output at half the
// instruction count of a one-pixel-per-lane kernel (the form rounds 1-5 kept for rectangular windows).
//
// Square windows: everything static (template N).  RECTANGULAR windows (reference test: 5 x 3, test/iterative/test_savgol2d.c:508-543): the geometry
// along x is the template's (N = half_window_x), the number of window ROWS is a run-time count wwy <= CW (two builds: CW = 9 and 33).  The chain of
// accumulator slots is anchored at its END: block s (s = CW-1 down to CW-wwy) feeds W row s - (CW - wwy), its first add takes slot s and writes slot
// s + 1, the others run in place there; slot CW is the finished sum, slot CW - wwy is never written and stays the +0 every output starts from, and the
// walk stops at the first block the filter does not have.  Same products, same adds, same order as the reference -- the same bits.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "sg_2d.hpp"
#include "sg_pk.hpp"
#include "sg_runtime.hpp"

namespace sg {

typedef f32x2 __attribute__((address_space(4))) ConstPair;

constexpr int DENSE_RECT_SHORT = 4;                         // rectangular windows with half_window_y <= 4: a chain of 9 slots instead of 33

template <int N>
struct Dense {
    static constexpr int WW = 2 * N + 1;
    static constexpr int WP = 2 * N + 2;                    // row pitch of the tap table the host uploads (pairs, zero padded)
    static constexpr int HL = (N + 3) / 4;                  // halo lanes on each side of a strip
    static constexpr int OUTL = 64 - 2 * HL;                // lanes whose columns are stored
    static constexpr int SW = 4 * OUTL;                     // stored columns per strip
    static constexpr int NQ = 2 * HL + 1;                   // 16-byte quads of the input row a lane reads back
    static constexpr int D = 4 * HL - N;                    // window index of the first tap of output 0
    static constexpr int P = 2;                             // input rows loaded ahead
    static constexpr int BUFW = 256 + 8 * HL;               // LDS floats per row (strip + pad both sides)
};

// w[SEL] * x like pk_mul_sgpr (sg_pk.hpp), but volatile, and so are the adds: left to the compiler, the (2N+1)^2
// products of a row are all issued ahead of the adds (or carried into the next loop iteration) and spill
template <int SEL>
__device__ __forceinline__ f32x2 pk_mul_here(const f32x2 wpair, const f32x2 x)
{
    f32x2 p;
    if constexpr (SEL == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "s"(wpair), "v"(x));
    else                    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(p) : "s"(wpair), "v"(x));
    return p;
}

__device__ __forceinline__ int dense_fix_row(int i, int n, bool reflect)      // fix_index of sg_2d.hpp, branch free (uniform)
{
    const int below = reflect ? ~i : 0;
    const int above = reflect ? 2 * n - 1 - i : n - 1;
    int a = i < 0 ? below : (i >= n ? above : i);
    a = a < 0 ? 0 : a;
    return a >= n ? n - 1 : a;
}

// One item: the strip of SW stored columns at sx, output rows yb .. yb+nout-1 of one frame (see sg_2d_roll.hip for the
// strip / band geometry; VEC = interior strip with 16-byte aligned rows).
// CAP = 0: square window, WW rows.  CAP > 0: wwy = 2 * half_window_y + 1 rows at run time (half_window_y <= CAP), N = half_window_x.
template <int N, bool VEC, int CAP = 0>
__device__ __forceinline__ void dense_item(const Job2D &job, const float *__restrict__ W, float *mine, const float *in, float *out,
                                           int sx, int yb, int nout, int lane, int xlo, int xhi, int ylo, int yhi)
{
    typedef Dense<N> R;
    constexpr bool RT = CAP > 0;
    constexpr int CW = RT ? 2 * CAP + 1 : R::WW;                   // window rows the chain is compiled for
    const int ny = RT ? job.ny : N, wwy = 2 * ny + 1, s0 = CW - wwy;
    const int c0 = sx - 4 * R::HL + 4 * lane;                // this lane's first column (frame coordinates)
    int ix0 = 0, ix1 = 0, ix2 = 0, ix3 = 0;
    if constexpr (!VEC) {
        ix0 = fix_index(c0, job.cols, job.boundary); ix1 = fix_index(c0 + 1, job.cols, job.boundary);
        ix2 = fix_index(c0 + 2, job.cols, job.boundary); ix3 = fix_index(c0 + 3, job.cols, job.boundary);
    }
    const bool reflect = job.boundary == SAVGOL2D_BOUNDARY_REFLECT;
    auto load_row = [&](int r) -> f32x4 {                    // band input row r = frame row yb-N+r, remapped at the border
        const float *row = in + (long long)dense_fix_row(yb - ny + r, job.rows, reflect) * job.in_stride;
        if constexpr (VEC) return *reinterpret_cast<const f32x4 *>(row + c0);
        else return f32x4{row[ix0], row[ix1], row[ix2], row[ix3]};
    };
    const bool out_lane = lane >= R::HL && lane < 64 - R::HL;
    float *const wr = mine + 4 * R::HL + 4 * lane;           // where this lane's 4 columns of the input row go
    const float *const rd = mine + 4 * lane;                 // where its window starts

    // acc[wy][j]: output row (r - wy), column pair j, while input row r is being fed.  Slots that belong to rows above
    // the band hold garbage until a real output row starts in slot 0; nothing of them is ever stored.
    // (run-time row count: one more slot, acc[CW] = the finished sum)
    f32x2 acc[CW + (RT ? 1 : 0)][2];
#pragma unroll
    for (int a = 0; a < CW + (RT ? 1 : 0); ++a) { acc[a][0] = f32x2{0.0f, 0.0f}; acc[a][1] = f32x2{0.0f, 0.0f}; }
    f32x4 ahead[R::P];
#pragma unroll
    for (int p = 0; p < R::P; ++p) ahead[p] = load_row(p);

    const int nrows = nout + 2 * ny;
    for (int r = 0; r < nrows; ++r) {
        // the row crosses lanes through LDS (two alternating rows: the next write never races this read)
        float *buf = mine + (r & 1) * R::BUFW;
        *reinterpret_cast<f32x4 *>(buf + (wr - mine)) = ahead[0];
#pragma unroll
        for (int p = 0; p + 1 < R::P; ++p) ahead[p] = ahead[p + 1];
        ahead[R::P - 1] = load_row(r + R::P);                // rows past the band are clamped re-reads, never used
        wave_lds_sync();
        f32x2 e[2 * R::NQ + 1];
#pragma unroll
        for (int q = 0; q < R::NQ; ++q) {
            const f32x4 w4 = *reinterpret_cast<const f32x4 *>(buf + (rd - mine) + 4 * q);
            e[2 * q] = f32x2{w4.x, w4.y};
            e[2 * q + 1] = f32x2{w4.z, w4.w};
        }
        f32x2 pr[2 * N + 3];                                 // pr[j] = window floats (D+j, D+j+1)
#pragma unroll
        for (int j = 0; j < 2 * N + 3; ++j) {
            const int idx = R::D + j;
            pr[j] = (idx & 1) ? pk_straddle(e[idx >> 1], e[(idx >> 1) + 1]) : e[idx >> 1];
        }
        // feed row r into every output row that sees it: sum += W[wy][wx] * in, wx ascending, separate roundings.
        // The taps come through scalar loads (uniform address, constant address space: the table is never written
        // while a kernel runs) as aligned pairs, one W row ahead of the arithmetic.  Each row's pointer is laundered
        // through an empty asm that also names an accumulator of the row before last: that pins the loads to this
        // place -- hoisted out of the loop, or all issued at the top of it, the (2N+1)^2 taps need 225 SGPRs at N = 7.
        f32x2 wcur[R::WP / 2], wnext[R::WP / 2];
        auto load_taps = [&](f32x2 (&dst)[R::WP / 2], int wy, const f32x2 after) {
            const float *wptr = W + wy * R::WP;
            asm volatile("" : "+s"(wptr) : "v"(after));
            const ConstPair *wrow = reinterpret_cast<const ConstPair *>(reinterpret_cast<uintptr_t>(wptr));
#pragma unroll
            for (int i = 0; i < R::WP / 2; ++i) dst[i] = wrow[i];
        };
        // Output rows are independent, so W's rows are walked from the last one down: the final add of row wy can then
        // write its sum straight into slot wy+1 (already drained), which is where the next input row expects it -- the
        // accumulators shift without a single move.  Slot 0 starts each output with 0 + product, as the reference does.
        f32x2 done0, done1;
        load_taps(wcur, wwy - 1, pr[0]);
        if constexpr (RT) {
            static_for<CW>([&](auto sc) -> bool {
                constexpr int s = CW - 1 - decltype(sc)::value;
                if (s < s0) return false;                          // uniform: the window rows this filter does not have come last in the walk
                const int wy = s - s0;
                load_taps(wnext, wy > 0 ? wy - 1 : 0, s + 2 <= CW ? acc[s + 2 <= CW ? s + 2 : 0][1] : pr[1]);
                f32x2 p0 = pk_mul_here<0>(wcur[0], pr[0]), p1 = pk_mul_here<0>(wcur[0], pr[2]);
                static_for<R::WW>([&](auto wxc) -> bool {
                    constexpr int wx = decltype(wxc)::value;
                    f32x2 n0 = p0, n1 = p1;
                    if constexpr (wx + 1 < R::WW) {
                        n0 = pk_mul_here<((wx + 1) & 1)>(wcur[(wx + 1) >> 1], pr[wx + 1]);
                        n1 = pk_mul_here<((wx + 1) & 1)>(wcur[(wx + 1) >> 1], pr[wx + 3]);
                    }
                    if constexpr (wx == 0) {                      // the sum so far (slot s; +0 for W's first row) moves on to slot s + 1
                        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc[s + 1][0]) : "v"(acc[s][0]), "v"(p0));
                        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc[s + 1][1]) : "v"(acc[s][1]), "v"(p1));
                    } else {
                        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[s + 1][0]) : "v"(p0));
                        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[s + 1][1]) : "v"(p1));
                    }
                    p0 = n0; p1 = n1;
                    return true;
                });
#pragma unroll
                for (int i = 0; i < R::WP / 2; ++i) wcur[i] = wnext[i];
                return true;
            });
            done0 = acc[CW][0]; done1 = acc[CW][1];
        } else {
        static_for<CW>([&](auto wyc) -> bool {
            constexpr int wy = CW - 1 - decltype(wyc)::value;
            constexpr bool last_slot = wy + 1 == CW;               // the chain's last sum goes to done0 / done1
            if constexpr (wy > 0) load_taps(wnext, wy - 1, !last_slot ? acc[!last_slot ? wy + 1 : 0][1] : pr[1]);   // (index clamped for the dead arm: -Warray-bounds)
            // the products of tap wx+1 are issued before the adds of tap wx (see sg_pk.hpp on asm results and s_nop)
            f32x2 p0 = pk_mul_here<0>(wcur[0], pr[0]), p1 = pk_mul_here<0>(wcur[0], pr[2]);
            static_for<R::WW>([&](auto wxc) -> bool {
                constexpr int wx = decltype(wxc)::value;
                f32x2 n0 = p0, n1 = p1;
                if constexpr (wx + 1 < R::WW) {
                    n0 = pk_mul_here<((wx + 1) & 1)>(wcur[(wx + 1) >> 1], pr[wx + 1]);
                    n1 = pk_mul_here<((wx + 1) & 1)>(wcur[(wx + 1) >> 1], pr[wx + 3]);
                }
                if constexpr (wx + 1 == R::WW) {              // last tap of this W row: the sum moves on to the next slot
                    f32x2 &d0 = !last_slot ? acc[!last_slot ? wy + 1 : 0][0] : done0;
                    f32x2 &d1 = !last_slot ? acc[!last_slot ? wy + 1 : 0][1] : done1;
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d0) : "v"(acc[wy][0]), "v"(p0));
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d1) : "v"(acc[wy][1]), "v"(p1));
                } else if constexpr (wy == 0 && wx == 0) {    // a new output row: sum = 0 + w * x
                    asm volatile("v_pk_add_f32 %0, %1, 0 op_sel_hi:[1,0]" : "=v"(acc[0][0]) : "v"(p0));
                    asm volatile("v_pk_add_f32 %0, %1, 0 op_sel_hi:[1,0]" : "=v"(acc[0][1]) : "v"(p1));
                } else {
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[wy][0]) : "v"(p0));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[wy][1]) : "v"(p1));
                }
                p0 = n0; p1 = n1;
                return true;
            });
#pragma unroll
            for (int i = 0; i < R::WP / 2; ++i) wcur[i] = wnext[i];
            return true;
        });
        }
        // output row r-2ny has seen its last input row
        const int yo = yb + r - 2 * ny;
        if (r >= 2 * ny && yo >= ylo && yo < yhi) {          // uniform
            const f32x2 s2 = f32x2{job.scale, job.scale};
            const f32x2 o0 = done0 * s2, o1 = done1 * s2;
            float *orow = out + (long long)yo * job.out_stride;
            if constexpr (VEC) {
                if (out_lane)
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, f32x4{o0.x, o0.y, o1.x, o1.y}), reinterpret_cast<u32x4 *>(orow + c0));
            } else if (out_lane) {
                if (c0 >= xlo && c0 < xhi) orow[c0] = o0.x;
                if (c0 + 1 >= xlo && c0 + 1 < xhi) orow[c0 + 1] = o0.y;
                if (c0 + 2 >= xlo && c0 + 2 < xhi) orow[c0 + 2] = o1.x;
                if (c0 + 3 >= xlo && c0 + 3 < xhi) orow[c0 + 3] = o1.y;
            }
        }
    }
    wave_lds_sync();                                         // the next item's first write must stay behind these reads
}

template <int N, int CAP>
__global__ __launch_bounds__(256) void sg2d_dense_roll_kernel(const Job2D job, const float *__restrict__ W, unsigned strips, unsigned bands,
                                                              int band_rows, unsigned total_items, int aligned)
{
    typedef Dense<N> R;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *mine = lds + wv * (2 * R::BUFW);                  // two LDS rows, private to this wave

    const unsigned nblk = gridDim.x;
    const unsigned blk = (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3);
    const unsigned nwaves = nblk * 4u;

    const bool valid = job.boundary == SAVGOL2D_BOUNDARY_VALID;
    const int xlo = valid ? N : 0, xhi = valid ? job.cols - N : job.cols;
    const int ny = CAP > 0 ? job.ny : N;
    const int ylo = valid ? ny : 0, yhi = valid ? job.rows - ny : job.rows;

    for (unsigned item = blk * 4u + (unsigned)wv; item < total_items; item += nwaves) {
        const unsigned strip = item % strips, ib = item / strips;
        const unsigned band = ib % bands, img = ib / bands;
        const int sx = (int)strip * R::SW, yb = (int)band * band_rows;
        const int nout = job.rows - yb < band_rows ? job.rows - yb : band_rows;
        const float *in = job.in + (long long)img * job.in_pitch;
        float *out = job.out + (long long)img * job.out_pitch;
        if (aligned == 3 && sx - 4 * R::HL >= 0 && sx - 4 * R::HL + 256 <= job.cols && sx >= xlo && sx + R::SW <= xhi)
            dense_item<N, true, CAP>(job, W, mine, in, out, sx, yb, nout, lane, xlo, xhi, ylo, yhi);
        else
            dense_item<N, false, CAP>(job, W, mine, in, out, sx, yb, nout, lane, xlo, xhi, ylo, yhi);
    }
}

template <int N, int CAP>
static int launch_dense(const Job2D &job, const float *d_w, unsigned images, int cu_count, hipStream_t st)
{
    typedef Dense<N> R;
    const unsigned strips = (unsigned)((job.cols + R::SW - 1) / R::SW);
    static int per_cu = 0;                                   // resident blocks per CU of this instantiation
    const size_t lds = sizeof(float) * 4 * 2 * R::BUFW;
    if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sg2d_dense_roll_kernel<N, CAP>, 256, lds) != hipSuccess || nb < 1) nb = 2;
        per_cu = nb > 4 ? 4 : nb;
    }
    const unsigned nwaves = (unsigned)cu_count * (unsigned)per_cu * 4u;
    unsigned bands = choose_bands(job.rows, (unsigned long long)images * strips, nwaves, CAP > 0 ? job.ny : N, 1.0);   // warm-up rows are fed in full
    const int band_rows = (int)((job.rows + (int)bands - 1) / (int)bands);
    const unsigned long long total = (unsigned long long)images * strips * bands;     // caller keeps this < 2^32
    unsigned grid = (unsigned)cu_count * (unsigned)per_cu;
    if ((unsigned long long)grid * 4ull > total) grid = (unsigned)((total + 3) / 4);
    grid = (grid + 7u) & ~7u;
    int aligned = 0;
    if (job.in_stride % 4 == 0 && job.in_pitch % 4 == 0 && (reinterpret_cast<uintptr_t>(job.in) & 15u) == 0) aligned |= 1;
    if (job.out_stride % 4 == 0 && job.out_pitch % 4 == 0 && (reinterpret_cast<uintptr_t>(job.out) & 15u) == 0) aligned |= 2;
    hipLaunchKernelGGL((sg2d_dense_roll_kernel<N, CAP>), dim3(grid), dim3(256), lds, st, job, d_w, strips, bands, band_rows, (unsigned)total, aligned);
    return 0;
}

template <int N>
static int dispatch_dense(int n, bool rect, const Job2D &job, const float *d_w, unsigned images, int cu_count, hipStream_t st)
{
    if (n == N) {
        if (!rect) return launch_dense<N, 0>(job, d_w, images, cu_count, st);
        // rectangular: the build whose accumulator chain is just long enough (the short one leaves registers for four waves per SIMD)
        return job.ny <= DENSE_RECT_SHORT ? launch_dense<N, DENSE_RECT_SHORT>(job, d_w, images, cu_count, st)
                                          : launch_dense<N, DENSE_ROLL_MAX_N>(job, d_w, images, cu_count, st);
    }
    if constexpr (N < DENSE_ROLL_MAX_N) return dispatch_dense<N + 1>(n, rect, job, d_w, images, cu_count, st);
    else return 1;
}

// 0 = launched, 1 = not covered (a half window > DENSE_ROLL_MAX_N: none that savgol2d_create accepts), -1 = error.
// h_w = the filter's [2ny+1][2nx+1] kernel on the host; it is uploaded once per distinct content with rows padded to an even
// number of floats, so that a row's taps are aligned pairs.
int sg2d_launch_dense_rolling(const Job2D &job, const float *h_w, DeviceCtx *ctx, unsigned images, hipStream_t st)
{
    const int nx = job.nx, ny = job.ny, ww = 2 * nx + 1, wp = 2 * nx + 2, wwy = 2 * ny + 1;
    if (nx < 1 || ny < 1 || nx > DENSE_ROLL_MAX_N || ny > DENSE_ROLL_MAX_N) return 1;
    float padded[(2 * DENSE_ROLL_MAX_N + 1) * (2 * DENSE_ROLL_MAX_N + 2)];
    memset(padded, 0, sizeof(padded));
    for (int wy = 0; wy < wwy; ++wy) memcpy(padded + wy * wp, h_w + wy * ww, sizeof(float) * ww);
    const float *d_w = ctx_table(ctx, padded, sizeof(float) * (size_t)wwy * wp, 0x2e000000u + (unsigned)(nx * 64 + ny));
    if (!d_w) return -1;
    return dispatch_dense<1>(nx, nx != ny, job, d_w, images, ctx->cu_count, st);
}

}  // namespace sg
