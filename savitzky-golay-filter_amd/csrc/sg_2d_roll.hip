// sg_2d_roll.hip -- the 2-D fast path for every half window (1..16): rolling column windows in registers.
//
// Same exact low-rank factorisation as sg_2d_sep.hip,  W(x,y) = sum_t G_t(y) Q_t(x)  (reference kernel:
// src/savgol2d.c:188-265), applied vertical pass first:
//      v_t = G_t (*)y in        out = sum_t Q_t (*)x v_t
// One WAVE owns a 256-column strip (a lane owns 4 adjacent columns = one 16-byte load per row) and walks
// down a band of rows.  The last 2N+1 input rows of its columns stay in registers (a ring that the fully
// unrolled loop indexes with literals), so the vertical pass costs no LDS traffic and no halo rows; only
// the r vertical results of the current row cross lanes, through a wave-private LDS row (no
// __syncthreads anywhere: LDS operations of one wave execute in order).  The 2*HL outermost lanes of a
// strip are halo (their columns are recomputed by the neighbouring strip): 224 to 240 of 256 columns are stored.
//
// Every term of one kernel has the same parity in y ((-1)^dy) and in x ((-1)^dx), so both passes fold
//      sum_k w[k] s[k]  =  sum_{k<N} w[k] (s[k] +- s[2N-k])  +  w[N] s[N]
// which halves the taps held in SGPRs (N+1 per vector, passed by value in the kernarg) and lets all terms
// share the folded vertical window.  Arithmetic is v_pk_fma_f32 throughout (two columns per instruction).
//
// Rounding differs from the reference's dense sum at the 1e-7 level, like the tile kernel of sg_2d_sep.hip
// (tests bound both at 1e-6 normwise of the double-accumulation oracle).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "sg_2d.hpp"
#include "sg_pk.hpp"
#include "sg_runtime.hpp"

namespace sg {

#ifdef SG_STAMPS2D   // diagnostic build only (tools/stamp_2d.py): s_memtime stamps of ONE wave's phases.  Never timed as a product number.
__device__ unsigned long long *g_stamps2d;
__device__ __forceinline__ void stamp2d(bool on, int slot)
{
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (on && slot < 512 && (threadIdx.x & 63) == 0) g_stamps2d[slot] = t;
}
#define SG_STAMP2D(slot) stamp2d(stamp_on && (slot) < 128, stamp_base + (slot))
#else
#define SG_STAMP2D(slot) do {} while (0)
#endif

// TR > 0: TILE form (round 4).  The item is TR output rows; ALL its TR + 2N input rows are loaded up front (P = TR - 1 rows "ahead"),
// the row loop runs once over literal slots and the wave exits: short-lived waves handed out in address order, the 1-D kernel's life
// cycle.  The 2N halo rows are read again by the tile below -- out of L2 when blocks that share an XCD take neighbouring tiles.
template <int N, bool ACC = false, int TR = 0>
struct Roll {
    // halo lanes on each side of a strip: ceil(N/4) are needed; more where that buys whole memory lines.  A strip stores
    // 1024 - 32*HL bytes per row starting 16*HL bytes into its loaded KiB: with HL = 1 (3) the stores of neighbouring strips meet
    // inside 64-byte halves of a line.  Copying frames with this walk and no arithmetic (tools/membench2d.hip) moves 4.72 TB/s
    // with HL = 1, 5.15 with HL = 2 (stores on 64-byte boundaries) and 5.22 with HL = 4 (stores on 128-byte lines); the kernel
    // itself gains 8-11 % at n <= 4 with HL = 2 and 7 % at n = 5, 6 with HL = 4; n = 7, 8 and the VALU-bound n >= 9 gain nothing
    // (tools/ab_2d.py, same process)
#ifdef SG_ROLL_HL_MIN
    static constexpr int HL = (N + 3) / 4 > SG_ROLL_HL_MIN ? (N + 3) / 4 : SG_ROLL_HL_MIN;
#else
    static constexpr int HL = N <= 4 ? 2 : (N <= 6 ? 4 : (N + 3) / 4);
#endif
    static constexpr int OUTL = 64 - 2 * HL;                // lanes whose columns are stored
    static constexpr int SW = 4 * OUTL;                     // stored columns per strip
    static constexpr int NQ = 2 * HL + 1;                   // 16-byte quads a lane reads back per term
    static constexpr int D = 4 * HL - N;                    // window index of the first tap of output 0
    // rows loaded ahead of the arithmetic (odd: U must be even).  PMC on config 4 (n=7) showed the waves parked on s_waitcnt 47 %
    // of their cycles with one row ahead; three rows ahead cost 8 VGPRs and buy 5-10 % from n = 6 up (n = 8: 2.21 -> 2.00 ms per
    // 64 frames, n = 9: 2.41 -> 2.29), nothing or a loss below (tools/ab_2d.py, all builds in one process)
    // (accumulating passes, n >= 9: the output rows they add to are prefetched through a ring of four slots, which wants U to be a
    //  multiple of four: P = 5 at odd half windows)
#ifdef SG_ROLL_P
    static constexpr int P = TR > 0 ? TR - 1 : SG_ROLL_P;
#else
    static constexpr int P = TR > 0 ? TR - 1 : ((ACC && (N & 1) && N < 15) ? 5 : (N >= 6 ? 3 : 1));
#endif
    static constexpr int U = 2 * N + 1 + P;                 // ring slots = unroll factor of the row loop
    // branch-free row loop (buffer stores whose range check replaces the `if`, whole groups of U rows without an exit test).
    // It paid (+4 % at n=7, +12-15 % at n = 10, 12) while the LDS reads of a row were issued right before their use: the branches
    // made hipcc wait for vmcnt(1) where vmcnt(6) was meant.  With the reads issued a pass early (see roll_item) the plain loop
    // is as fast or faster at every half window (n=7: 1.96 vs 2.04 ms, n=12: 2.87 vs 3.04), so it is off; the macro keeps the
    // variant buildable for the next compiler.
#ifdef SG_ROLL_STRAIGHT
    static constexpr bool STRAIGHT = TR > 0 || SG_ROLL_STRAIGHT != 0;
#else
    static constexpr bool STRAIGHT = TR > 0;                // a tile's waits must stay counted: 2N + TR loads are in flight when its first row starts
#endif
    static constexpr int BUFW = 256 + 8 * HL;               // LDS floats per term row (strip + pad both sides)
    static constexpr int NP = N / 2 + 1;                    // SGPR pairs holding taps 0..N
};

// taps 0..N of every vector (the mirrored half follows from the parity), Q already multiplied by the output scale
// NOUT output frames share one walk over the input (gradient: d/dx and d/dy); each has its own NT terms and parities
template <int N, int NT, int NOUT>
struct RollTaps {
    f32x2 g[NOUT][NT][Roll<N>::NP];
    f32x2 q[NOUT][NT][Roll<N>::NP];
    f32x2 sy[NOUT], sx[NOUT];                               // +1 / -1 (both halves equal)
    f32x2 qx[2 * N + 2];                                     // additive form, x-stationary horizontal pass: (a[k], a[k-1]), k = 0 .. 2N+1, a[-1] = a[2N+1] = 0
};

// The additive form's weighted horizontal unit, X-STATIONARY (round 5): one window float, broadcast, times the SGPR pair (a[k], a[k-1]) feeds the
// output pair (c, c + 1) -- no folded pairs, no v_pk_mov_b32 for the pairs that straddle two aligned registers: 2 (2N + 2) multiply-adds per lane and
// row instead of 2 (N + 1) + 2 N folds + ~N moves (n = 7: 32 against 39 of the row's 94 vector instructions).  Same idiom as sg_k1d_momenth.hpp.
#ifndef SG_ROLL_XST
#define SG_ROLL_XST 1
#endif
template <int HALF>
__device__ __forceinline__ void roll_fma_xb(f32x2 &acc, const f32x2 s, const f32x2 x)
{
    if constexpr (HALF == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(s), "v"(x));
    else                     asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(s), "v"(x));
}
template <int HALF>
__device__ __forceinline__ f32x2 roll_mul_xb(const f32x2 s, const f32x2 x)
{
    f32x2 p;
    if constexpr (HALF == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "s"(s), "v"(x));
    else                     asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(p) : "s"(s), "v"(x));
    return p;
}

// One item: a 256-column strip whose first loaded column is xload, output rows yb .. yb+nout-1 of one frame.  VEC: the strip's 256
// input columns are inside the frame, all its SW output columns are stored and rows are 16-byte aligned (one
// dwordx4 load and store per lane per row); otherwise four remapped scalar loads and masked scalar stores (the
// strips at the left / right frame edge, odd strides).
// (Round 3 tried frame-edge strips on vector loads as well -- the columns beyond the edge synthesised in the LDS row from the
// lane's own vertical results, 248 columns stored per edge strip, so that a 4096-column frame is 2 x 248 + 15 x 240 = 17 strips
// instead of 18.  It lost 6-17 % at every width tried (3840 ... 4320 columns, n = 2 ... 12): the inner strips then start at
// 248 + 240 k, and stores of neighbouring strips that meet inside a 64-byte half line are exactly what tools/membench2d.hip had
// found to cost 7-10 %.  Removed again; profiles/r03_2d_experiments.txt has the numbers.)
// Row r of the band's input (frame row yb-N+r, remapped at the frame border) lives in ring slot r % U.  Output row m
// needs rows m .. m+2N; while it is computed, row m+2N+P is already being loaded into the slot row m-1 left.
// The two passes are skewed by one row: iteration m runs the vertical pass of row m and writes its results to one LDS
// row, then reads row m-1's horizontal window from the other LDS row (written one iteration earlier, so the data
// is there when the reads issue) and does row m-1's horizontal arithmetic and store.
//
// BOX (NT = 2, NOUT = 1): the kernel is ADDITIVE, W(x, y) = A(x) + B(y) with B(0) = 0 -- every smoothing kernel of order <= 3
// (BASELINE config 4) is: only 1, x^2, y^2 survive the window's symmetry.  Then
//      out = A (*)x box_y(in)  +  box_x( B (*)y in )
// and the two box factors are not multiply-add passes: box_y is a rolling sum down the ring (add the row that enters, subtract
// the row that leaves; re-seeded from the ring every U rows, so the drift is bounded by U steps), box_x over a lane's four
// outputs is one sum of the aligned pairs they share plus a sliding correction.  89 instead of 125 VALU instructions per row
// of 256 columns at n = 7.
//
// ACC (NOUT = 1): the result is ADDED to what the output frame holds -- the second launch of a kernel whose terms do not fit one
// (taps live in SGPRs: 4 terms at n >= 9, 3 at n >= 13).  The stored row's previous content is loaded one row step ahead.
// MODE: 0 = remapped scalar loads and masked scalar stores (any frame); 1 = the strip's 256 columns are inside the frame and every stored
// quad is whole (one dwordx4 load and store per lane and row); 2, 3 = round 4, tile form only: a strip that reaches over the LEFT / RIGHT
// frame edge on vector loads too.  A lane's quad of columns is either inside the frame or outside it (16-byte aligned rows, cols % 4 == 0);
// a lane outside loads the quad its columns map to (reference src/savgol2d.c:428-445: REFLECT = the mirrored quad, reversed; CONSTANT =
// the frame's first / last quad, its edge float broadcast) and fixes the order with selects -- 6 v_cndmask per row, no scalar loads, no
// branch.  2: padded modes, stored quads are whole.  3: VALID, where the stored range starts and ends inside a quad: four range-checked
// dword stores per row for the lanes that hold such a quad.  (The edge strips on the scalar path cost the tile form 13 %:
// profiles/r04_2d_tile_experiments.txt.)
template <int N, int NT, int NOUT, int MODE, bool BOX, bool ACC, int TR = 0>
__device__ __forceinline__ void roll_item(const Job2D &job, const RollTaps<N, NT, NOUT> &taps, float *mine, const float *in, float *const (&outs)[NOUT],
                                          int xload, int yb, int nout, int lane, int xlo, int xhi, int ylo, int yhi)
{
    typedef Roll<N, ACC, TR> R;
    static_assert(!BOX || (NT == 2 && NOUT == 1), "the additive form is one output of two terms");
    static_assert(TR == 0 || !ACC, "tiles are plain passes");
    static_assert(!ACC || (NOUT == 1 && !BOX && !R::STRAIGHT), "accumulating passes are single-output launches of the general form");
    static_assert(R::U % 2 == 0, "the LDS row alternates with the ring slot: U must be even");
    constexpr bool VEC = MODE != 0, EDGE = MODE >= 2, PARTIAL = MODE == 3;
    static_assert(!EDGE || R::STRAIGHT, "edge strips on vector loads store through the range-checked descriptor");
    const int c0 = xload + 4 * lane;                         // this lane's first column (frame coordinates)
    int qcol = c0;                                           // EDGE: the in-frame quad this lane loads ...
    bool fx_w = false, fx_x = false, fx_rev = false, fx_b = false;      // ... and how its floats are permuted
    if constexpr (EDGE) {
        const bool left = c0 < 0, right = c0 >= job.cols;
        if (job.boundary == SAVGOL2D_BOUNDARY_REFLECT) {
            qcol = left ? -c0 - 4 : (right ? 2 * job.cols - c0 - 4 : c0);
            fx_rev = left || right;
        } else if (job.boundary == SAVGOL2D_BOUNDARY_CONSTANT) {
            fx_b = left || right;
        }
        qcol = qcol < 0 ? 0 : (qcol > job.cols - 4 ? job.cols - 4 : qcol);      // CONSTANT, VALID, and lanes further out than anything stored needs
        fx_w = fx_rev || (fx_b && right);                    // .x comes from .w (mirrored, or the frame's last float broadcast)
        fx_x = fx_rev || (fx_b && left);                     // .w comes from .x
    }
    int ix0 = 0, ix1 = 0, ix2 = 0, ix3 = 0;
    if constexpr (!VEC) {
        ix0 = fix_index(c0, job.cols, job.boundary); ix1 = fix_index(c0 + 1, job.cols, job.boundary);
        ix2 = fix_index(c0 + 2, job.cols, job.boundary); ix3 = fix_index(c0 + 3, job.cols, job.boundary);
    }
    const bool reflect = job.boundary == SAVGOL2D_BOUNDARY_REFLECT;
    auto load_row = [&](int r) -> f32x4 {                    // rows past the band are clamped re-reads that are never used
        const float *row = in + (long long)fix_row(yb - N + r, job.rows, reflect) * job.in_stride;
#ifdef SG_ROLL_NT_LOADS                                           // A/B builds: streaming loads (the strip's halo columns then miss L2 for the neighbour)
        if constexpr (VEC) return __builtin_bit_cast(f32x4, __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(row + c0)));
#else
        if constexpr (EDGE) {
            const f32x4 q = *reinterpret_cast<const f32x4 *>(row + qcol);
            const float nx = fx_w ? q.w : q.x, nw = fx_x ? q.x : q.w;          // a broadcast lane has nx == nw == the edge float
            return f32x4{nx, fx_b ? nx : (fx_rev ? q.z : q.y), fx_b ? nx : (fx_rev ? q.y : q.z), nw};
        } else if constexpr (VEC) return *reinterpret_cast<const f32x4 *>(row + c0);
#endif
        else return f32x4{row[ix0], row[ix1], row[ix2], row[ix3]};
    };
    const bool out_lane = lane >= R::HL && lane < 64 - R::HL;
    const int yend = yb + nout;                              // first frame row past this band
    __amdgpu_buffer_rsrc_t rsrc[NOUT];                       // VEC stores go through a buffer descriptor per output frame (range-checked)
    const __amdgpu_buffer_rsrc_t rsrc_none = __builtin_amdgcn_make_buffer_rsrc(outs[0], 0, 0, 0x00020000);     // zero records: drops every store
    // EDGE: a lane stores its quad when all of it lies in the stored range; PARTIAL: the floats of a quad that straddles the range's end
    const bool whole = !EDGE || (c0 >= xlo && c0 + 4 <= xhi);
    const unsigned col_off = (out_lane && whole) ? (unsigned)(c0 * 4) : 0x80000000u;
    // (their four dword stores go through a per-row descriptor that spans exactly the row's stored range [xlo, xhi): the hardware range
    //  check drops the floats outside it.  One offset register per float, each laundered: left to see that the four offsets are
    //  consecutive, hipcc merges the four dword stores into ONE dwordx4 store, whose range check then passes or fails as a whole --
    //  the frame's border got written, and test_config4_full_size_frames caught it)
    unsigned pcol_off[4] = {0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u};
    if constexpr (PARTIAL) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (out_lane && !whole) pcol_off[j] = (unsigned)((c0 - xlo + j) * 4);
            asm volatile("" : "+v"(pcol_off[j]));
        }
    }
    if constexpr (VEC && R::STRAIGHT) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
            rsrc[o] = __builtin_amdgcn_make_buffer_rsrc(outs[o], 0, (int)((long long)job.rows * job.out_stride * 4), 0x00020000);
    }
    float *const wr = mine + 4 * R::HL + 4 * lane;           // where this lane's vertical results go
    const float *const rd = mine + 4 * lane;                 // where its horizontal window starts

    f32x4 win[R::U];

    f32x2 vbp[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};    // BOX: box sum of the previous output row minus its oldest input row
    // vertical pass of the output row whose first input row sits in slot u0 -> LDS row `par` (one row per output and term)
    auto vertical = [&](auto u0c, int par) {
        constexpr int u0 = decltype(u0c)::value;
        if constexpr (BOX) {
            // term 0, G = 1: the rolling box sum.  Slot u0 = 0 comes round once per U rows: re-seed from the ring there.
            auto lo2 = [](const f32x4 q) { return f32x2{q.x, q.y}; };
            auto hi2 = [](const f32x4 q) { return f32x2{q.z, q.w}; };
            f32x2 vb0, vb1;
            if constexpr (u0 == 0) {
                vb0 = lo2(win[0]); vb1 = hi2(win[0]);
#pragma unroll
                for (int k = 1; k <= 2 * N; ++k) { vb0 = vb0 + lo2(win[k]); vb1 = vb1 + hi2(win[k]); }
            } else {
                vb0 = vbp[0] + lo2(win[(u0 + 2 * N) % R::U]);
                vb1 = vbp[1] + hi2(win[(u0 + 2 * N) % R::U]);
            }
            vbp[0] = vb0 - lo2(win[u0]);                     // the row the next output row no longer sees
            vbp[1] = vb1 - hi2(win[u0]);
            const f32x4 vb = f32x4{vb0.x, vb0.y, vb1.x, vb1.y};
            // term 1, B(y) with B(0) = 0: N folds and N multiply-adds per column pair
            f32x2 v1[2], f[2][N];
            auto fold = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const f32x4 a = win[(u0 + k) % R::U], b = win[(u0 + 2 * N - k) % R::U];
                f[0][k] = lo2(a) + lo2(b);
                f[1][k] = hi2(a) + hi2(b);
            };
            fold(std::integral_constant<int, 0>{});
            static_for(std::make_integer_sequence<int, N>{}, [&](auto kc) -> bool {
                constexpr int k = decltype(kc)::value;
                if constexpr (k + 1 < N) fold(std::integral_constant<int, k + 1>{});
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if constexpr (k == 0) v1[c] = pk_mul_sgpr<0>(taps.g[0][1][0], f[c][0]);
                    else pk_fma_sgpr<(k & 1)>(v1[c], taps.g[0][1][k >> 1], f[c][k]);
                }
                return true;
            });
            float *row0 = mine + (par * 2 + 0) * R::BUFW, *row1 = mine + (par * 2 + 1) * R::BUFW;
            const f32x4 q1 = f32x4{v1[0].x, v1[0].y, v1[1].x, v1[1].y};
            *reinterpret_cast<f32x4 *>(row0 + 4 * R::HL + 4 * lane) = vb;
            *reinterpret_cast<f32x4 *>(row1 + 4 * R::HL + 4 * lane) = q1;
            return;
        }
        // Instruction order matters: the assembler pads an inline-asm result that is consumed within the next two
        // instructions with s_nop, so the fold of tap k+1 is issued before the multiply-adds of tap k and the
        // accumulator chains (2 column pairs x NT terms) are interleaved.
        static_for<NOUT>([&](auto oc) -> bool {
            constexpr int o = decltype(oc)::value;
            f32x2 v[NT][2], f[2][N + 1];
            auto fold = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const f32x4 a = win[(u0 + k) % R::U], b = win[(u0 + 2 * N - k) % R::U];
                if constexpr (k < N) {
                    f[0][k] = pk_fold(taps.sy[o], f32x2{b.x, b.y}, f32x2{a.x, a.y});
                    f[1][k] = pk_fold(taps.sy[o], f32x2{b.z, b.w}, f32x2{a.z, a.w});
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
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        if constexpr (k == 0) v[t][c] = pk_mul_sgpr<0>(taps.g[o][t][0], f[c][0]);
                        else pk_fma_sgpr<(k & 1)>(v[t][c], taps.g[o][t][k >> 1], f[c][k]);
                    }
                return true;
            });
#pragma unroll
            for (int t = 0; t < NT; ++t)
                *reinterpret_cast<f32x4 *>(wr + ((par * NOUT + o) * NT + t) * R::BUFW) = f32x4{v[t][0].x, v[t][0].y, v[t][1].x, v[t][1].y};
            return true;
        });
    };
    // horizontal pass in units of one (output, term): the unit's window comes out of LDS row `par` into one of two register
    // buffers, so the reads of unit u+1 are in flight while unit u is computed -- and the reads of a row's FIRST unit are issued
    // before the vertical pass of the next row (they depend on nothing in it), which hides the LDS round trip behind that
    // pass's 14 + 16 NT independent multiply-adds.  (Issued after it, as the first version did, every row had the wave stall
    // on lgkmcnt three to four times with nothing to do.)
    constexpr int UNITS = NOUT * NT;
    // (n = 15, 16: one buffer and the reads right before their use -- a second 36-register window does not fit beside the row ring;
    //  the additive form at n = 6, 7, 9, 10: its box unit consumes a window in ~16 instructions, too few to hide a second buffer's
    //  reads behind -- one buffer is 10-14 % faster at n = 6, 9, 10 and 0.5-3 % at n = 7, 2-4 % slower at n = 3, 5, 12, 14, level
    //  elsewhere: previous build vs this one in one process, profiles/r03_2d_experiments.txt exp12 / exp13)
    constexpr int NB = (N >= 15 || (BOX && (N == 6 || N == 7 || N == 9 || N == 10))) ? 1 : 2;
    f32x4 hq[NB][R::NQ];
    auto fetch = [&](auto uc, int par) {
        constexpr int u = decltype(uc)::value, o = u / NT, t = u % NT;
#pragma unroll
        for (int q = 0; q < R::NQ; ++q) hq[u & (NB - 1)][q] = *reinterpret_cast<const f32x4 *>(rd + ((par * NOUT + o) * NT + t) * R::BUFW + 4 * q);
    };
    // one unit's arithmetic on its fetched window, accumulated into r (the first term of an output starts it)
    auto hterm = [&](auto uc, f32x2 (&r)[2]) {
        constexpr int u = decltype(uc)::value, o = u / NT, t = u % NT;
        f32x2 e[2 * R::NQ + 1];
#pragma unroll
        for (int q = 0; q < R::NQ; ++q) {
            e[2 * q] = f32x2{hq[u & (NB - 1)][q].x, hq[u & (NB - 1)][q].y};
            e[2 * q + 1] = f32x2{hq[u & (NB - 1)][q].z, hq[u & (NB - 1)][q].w};
        }
        if constexpr (BOX && t == 1) {
            // Q = 1: box sums of the lane's four outputs.  Output j sums window floats lo+j .. hi+j; P = the aligned pairs inside
            // [lo, hi] (one packed add each, then the two halves), the rest is a sliding correction of single floats.
            constexpr int lo = R::D, hi = R::D + 2 * N;
            constexpr int plo = (lo + 1) / 2 * 2, phi = (hi + 1) / 2 * 2;            // aligned pairs plo, plo+2, .. < phi
            auto wf = [&](int i) -> float { return (i & 1) ? e[i >> 1].y : e[i >> 1].x; };
            f32x2 ps = e[plo >> 1];
#pragma unroll
            for (int i = plo + 2; i + 1 < phi; i += 2) ps = ps + e[i >> 1];
            float s0 = ps.x + ps.y;
            if constexpr (lo < plo) s0 += wf(lo);
            if constexpr (hi >= phi) s0 += wf(hi);
            const float s1 = s0 + (wf(hi + 1) - wf(lo));
            const float s2 = s1 + (wf(hi + 2) - wf(lo + 1));
            const float s3 = s2 + (wf(hi + 3) - wf(lo + 2));
            r[0] = r[0] + f32x2{s0, s1};
            r[1] = r[1] + f32x2{s2, s3};
            return;
        }
        if constexpr (SG_ROLL_XST && BOX && NOUT == 1 && t == 0 && TR > 0) {     // tiles only: on the strip walk (n >= 11) the 2N + 2 SGPR pairs cost more than the moves (4-9 % slower)
            // window float D + i is tap i - p of output c0 + p and tap i - p - 1 of output c0 + p + 1 (p = 0: r[0], p = 2: r[1]); two chains per
            // output pair (even / odd i), joined below
            f32x2 ch[2][2];
            static_for<2 * N + 4>([&](auto ic) -> bool {
                constexpr int i = decltype(ic)::value, idx = R::D + i;
                static_for<2>([&](auto pc) -> bool {
                    constexpr int p = 2 * decltype(pc)::value, kk = i - p;
                    if constexpr (kk >= 0 && kk <= 2 * N + 1) {
                        if constexpr (kk < 2) ch[p / 2][kk & 1] = roll_mul_xb<(idx & 1)>(taps.qx[kk], e[idx >> 1]);
                        else roll_fma_xb<(idx & 1)>(ch[p / 2][kk & 1], taps.qx[kk], e[idx >> 1]);
                    }
                    return true;
                });
                return true;
            });
            r[0] = ch[0][0] + ch[0][1];
            r[1] = ch[1][0] + ch[1][1];
            return;
        }
        f32x2 pr[2 * N + 3];                                 // pr[j] = window floats (D+j, D+j+1)
#pragma unroll
        for (int j = 0; j < 2 * N + 3; ++j) {
            const int idx = R::D + j;
            // e[2q], e[2q+1] are the halves of one read
            pr[j] = (idx & 1) ? (((idx >> 1) & 1) ? pk_straddle(e[idx >> 1], e[(idx >> 1) + 1]) : pk_middle(e[idx >> 1], e[(idx >> 1) + 1])) : e[idx >> 1];
        }
        f32x2 f[2][N + 1];
        auto fold = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k < N) {
                f[0][k] = pk_fold(taps.sx[o], pr[2 * N - k], pr[k]);
                f[1][k] = pk_fold(taps.sx[o], pr[2 + 2 * N - k], pr[2 + k]);
            } else {
                f[0][N] = pr[N];
                f[1][N] = pr[2 + N];
            }
        };
        fold(std::integral_constant<int, 0>{});
        static_for(std::make_integer_sequence<int, N + 1>{}, [&](auto kc) -> bool {
            constexpr int k = decltype(kc)::value;
            if constexpr (k < N) fold(std::integral_constant<int, k + 1>{});
            if constexpr (k == 0 && t == 0) { r[0] = pk_mul_sgpr<0>(taps.q[o][0][0], f[0][0]); r[1] = pk_mul_sgpr<0>(taps.q[o][0][0], f[1][0]); }
            else { pk_fma_sgpr<(k & 1)>(r[0], taps.q[o][t][k >> 1], f[0][k]); pk_fma_sgpr<(k & 1)>(r[1], taps.q[o][t][k >> 1], f[1][k]); }
            return true;
        });
    };
    // ACC: what the output frame holds in this lane's columns of frame row yo (rows past the frame: clamped, never stored)
    // Loaded as far ahead as the input rows (P row steps): the wait counter is in order, so a load consumed one step after its issue
    // would make every input row issued before it arrive within that step too -- the input prefetch would be worth one row, not P
    // (first version: the accumulating pass took 0.91 ms where the plain one takes 0.53).  Ring of P+1 slots indexed by literals.
    // (n = 15, 16: no room for four slots beside the 34-row ring -- two, loaded one step ahead)
    constexpr int PR = N >= 15 ? 2 : 4, LEAD = PR - 1;
    static_assert(!ACC || (R::U % PR == 0 && R::P >= LEAD), "slot = row % PR must carry over from one group of U rows to the next");
    f32x4 prevq[PR];
    auto load_prev = [&](int yo) -> f32x4 {
        const float *orow = outs[0] + (long long)(yo < job.rows ? yo : job.rows - 1) * job.out_stride;
        if constexpr (VEC) return *reinterpret_cast<const f32x4 *>(orow + c0);
        else {
            const bool row_ok = yo >= ylo && yo < yhi;
            auto at = [&](int c) -> float { return (row_ok && out_lane && c >= xlo && c < xhi) ? orow[c] : 0.0f; };
            return f32x4{at(c0), at(c0 + 1), at(c0 + 2), at(c0 + 3)};
        }
    };
    // the store of frame row yo of output `o`
    auto store_row = [&](auto oc, const f32x2 (&r)[2], int yo) {
        constexpr int o = decltype(oc)::value;
        if constexpr (VEC && R::STRAIGHT) {
            // ONE unconditional store instruction per row: a lane that must not store (strip halo, rows outside the band or the
            // stored range) gets an offset beyond the buffer and the hardware range check drops it.  A store under an `if` is a
            // branch, and behind a branch hipcc no longer knows how many memory operations are in flight: it then waits for
            // vmcnt(1) where vmcnt(6) would do, which drains the row prefetch (PMC: waves parked 41 % of their cycles).
            // The lane part of the offset (column bytes, or 2 GiB for a halo lane: beyond any frame the host lets through, with
            // or without the row part added) is fixed for the item; the row part is wave-uniform: one v_add per row.  A row
            // nobody stores selects the empty descriptor (scalar select).
            const bool keep_row = yo >= ylo && yo < yhi && yo < yend;                     // uniform
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{r[0].x, r[0].y, r[1].x, r[1].y}), keep_row ? rsrc[o] : rsrc_none,
                                                   (int)(col_off + (unsigned)(yo * job.out_stride * 4)), 0, 2 /* nt */);
            if constexpr (PARTIAL) {
                const float v[4] = {r[0].x, r[0].y, r[1].x, r[1].y};
                const __amdgpu_buffer_rsrc_t prow = __builtin_amdgcn_make_buffer_rsrc(outs[o] + (long long)yo * job.out_stride + xlo, 0,
                                                                                      keep_row ? (xhi - xlo) * 4 : 0, 0x00020000);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[j]), prow, (int)pcol_off[j], 0, 0);
            }
        } else if (yo >= ylo && yo < yhi && yo < yend) {     // uniform
            float *orow = outs[o] + (long long)yo * job.out_stride;
            if constexpr (VEC) {
                if (out_lane)
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, f32x4{r[0].x, r[0].y, r[1].x, r[1].y}),
                                                reinterpret_cast<u32x4 *>(orow + c0));
            } else if (out_lane) {
                if (c0 >= xlo && c0 < xhi) orow[c0] = r[0].x;
                if (c0 + 1 >= xlo && c0 + 1 < xhi) orow[c0 + 1] = r[0].y;
                if (c0 + 2 >= xlo && c0 + 2 < xhi) orow[c0 + 2] = r[1].x;
                if (c0 + 3 >= xlo && c0 + 3 < xhi) orow[c0 + 3] = r[1].y;
            }
        }
    };
    // units 0 .. UNITS-1 of the row in LDS row `par` (unit 0 already fetched): fetch the next unit, compute this one
    auto finish_row = [&](int par, int yo, const f32x4 prev) {
        f32x2 r[2];
        static_for<UNITS>([&](auto uc) -> bool {
            constexpr int u = decltype(uc)::value;
            if constexpr (NB == 1) fetch(uc, par);
            else if constexpr (u + 1 < UNITS) fetch(std::integral_constant<int, u + 1>{}, par);
            // the row's last reads are issued: order them before the next iteration's writes into the same LDS row.  Without a
            // branch between the iterations the compiler is free to hoist a lane's next write above reads of OTHER columns that it
            // can prove distinct for that lane -- which are exactly the words its neighbours are about to read
            if constexpr (u + 1 == UNITS) wave_lds_sync();
            // keep the reads ahead of this unit's arithmetic and that arithmetic ahead of the next unit's: left alone, the
            // scheduler pulls the first consumers of a read (the pair shuffles) up to right behind it and waits there
            __builtin_amdgcn_sched_barrier(0);
            hterm(uc, r);
            if constexpr (ACC && u == UNITS - 1) { r[0] = r[0] + f32x2{prev.x, prev.y}; r[1] = r[1] + f32x2{prev.z, prev.w}; }
            if constexpr (u % NT == NT - 1) store_row(std::integral_constant<int, u / NT>{}, r, yo);
            __builtin_amdgcn_sched_barrier(0);
            return true;
        });
    };

#ifdef SG_STAMPS2D
    // four candidate waves mid-launch (blocks 8 apart: one XCD), each with its own 128 slots; the tool reads the first that was an interior strip
    const unsigned stamp_c = (blockIdx.x - (gridDim.x / 2 & ~7u)) >> 3;
    const bool stamp_on = MODE == 1 && (blockIdx.x & 7u) == 0 && blockIdx.x >= (gridDim.x / 2 & ~7u) && stamp_c < 4 && (threadIdx.x >> 6) == 0;
    const int stamp_base = (int)(stamp_c & 3u) * 128;
#endif
    SG_STAMP2D(0);
#pragma unroll
    for (int r = 0; r < R::U; ++r) win[r] = load_row(r);         // rows 0..2N for the first output row, P more in flight
    SG_STAMP2D(1);                                               // all the item's first loads are issued
    vertical(std::integral_constant<int, 0>{}, 0);
    SG_STAMP2D(2);                                               // rows 0..2N have arrived and row 0's vertical pass is done
    if constexpr (ACC) {
#pragma unroll
        for (int j = 0; j < LEAD; ++j) prevq[j] = load_prev(yb + j);
    }
    if constexpr (TR > 0) {
        // the tile: every row is in flight already.  Step uu = vertical pass of row uu + 1, horizontal pass + store of row uu; rows past
        // the frame (nout < TR) are computed from clamped re-reads and dropped by the store's descriptor.  No branch in here.
        static_for(std::make_integer_sequence<int, TR>{}, [&](auto uuc) -> bool {
            constexpr int uu = decltype(uuc)::value;
            wave_lds_sync();
            if constexpr (NB == 2) fetch(std::integral_constant<int, 0>{}, uu & 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (uu + 1 < TR) vertical(std::integral_constant<int, uu + 1>{}, (uu + 1) & 1);
            SG_STAMP2D(3 + 2 * uu);                              // row uu+1's input has arrived, its vertical pass is done
            __builtin_amdgcn_sched_barrier(0);
            finish_row(uu & 1, yb + uu, f32x4{0.0f, 0.0f, 0.0f, 0.0f});
            SG_STAMP2D(4 + 2 * uu);                              // row uu's horizontal pass is done, its store issued
            return true;
        });
        return;
    }
    // Iteration m runs the vertical pass of row m and the horizontal pass + store of row m-1.  Whole groups of U iterations, no
    // early exit: the iterations past the band (at most U-1, their loads clamped to real rows) compute rows nobody stores.  An
    // exit test per row would be a branch per row, with the same cost to the wait counts as a branch around the store.
    for (int base = 1; base <= nout; base += R::U) {
        static_for(std::make_integer_sequence<int, R::U>{}, [&](auto uuc) -> bool {
            constexpr int uu = decltype(uuc)::value;
            const int m = base + uu;                         // base = 1 mod U: row m starts in slot (uu+1) % U
            if constexpr (!R::STRAIGHT) { if (m > nout) return false; }     // uniform; iteration m = nout still stores row nout-1
            win[uu] = load_row(m + R::U - 1);                // slot of row m-1, which no later row needs
            if constexpr (ACC) prevq[(uu + LEAD) % PR] = load_prev(yb + m - 1 + LEAD);       // for the store LEAD iterations on
            wave_lds_sync();                                 // row m-1's vertical results (previous iteration) are written ...
            if constexpr (NB == 2) fetch(std::integral_constant<int, 0>{}, uu & 1);     // ... and its first window is on its way while row m's vertical pass runs
            __builtin_amdgcn_sched_barrier(0);
            vertical(std::integral_constant<int, (uu + 1) % R::U>{}, (uu + 1) & 1);
            SG_STAMP2D(3 + 2 * (m - 1));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ACC) finish_row(uu & 1, yb + m - 1, prevq[uu % PR]);
            else finish_row(uu & 1, yb + m - 1, f32x4{0.0f, 0.0f, 0.0f, 0.0f});
            SG_STAMP2D(4 + 2 * (m - 1));
            return true;
        });
    }
}

// terms per output the rolling kernel is built for: the taps live in SGPRs (2 * NT * NOUT * (N/2 + 1) pairs), which caps the
// windows 9..12 at 3 terms and 13..16 at 2 (1 when two outputs share the walk); everything else runs the tile kernel of sg_2d_sep.hip
// (three outputs -- the Hessian -- share the walk up to n = 9 with one term each: every Hessian frame of order <= 3 is rank 1)
constexpr int roll_max_terms(int n, int nout) { return nout == 1 ? (n <= 8 ? SEP_MAX_TERMS : (n <= 12 ? 3 : 2)) : nout == 2 ? (n <= 8 ? 3 : 1) : (n <= 3 ? 2 : (n <= 9 ? 1 : 0)); }
// waves per SIMD the register allocation must allow: the row ring alone is (2N+2) x 4 VGPRs
// (n = 7 and n = 8 with three rows in flight and the two-output form at n = 6, rank 2, spill at 4 waves per SIMD: 300 / 20-108 / 12
// bytes of scratch; tools/roll_resources.py lists every instantiation)
#ifdef SG_ROLL_MINWAVES
constexpr int roll_min_waves(int, int, int) { return SG_ROLL_MINWAVES; }
#else
constexpr int roll_min_waves(int n, int nt, int nout) { return n >= 9 ? 2 : ((nt >= 3 && n >= 6) || n == 7 || n == 8 || (n == 6 && nt == 2 && nout == 2) || (nout == 3 && n >= 5) ? 3 : 4); }
#endif

// TILE form: output rows per tile (0 = this half window keeps the strip walk) and the waves per SIMD its registers must allow.
// tools/membench_tile2d.hip (profiles/r04_membench_tile2d.txt): the bare tile pattern moves 0.71-0.72 of the roofline at full
// occupancy whatever the tile height (8 ... 32 rows) and MORE on fewer resident waves (8 per CU: 0.757, 4 per CU: 0.775; the flat copy
// 0.80) -- so the row registers of a 16-row tile (30 rows x 4 VGPRs) cost nothing that matters.
// 20 rows since the end of round 5: once the x-stationary horizontal unit had freed its 2N + 3 window pairs the taller tile fits the same three waves per
// SIMD, and over fresh buffer pairs in one process (tools/placement_2d.py, profiles/r05_2d_tile_rows.txt) 20-row tiles are 1.4-2.4 % faster than 16 at
// n = 2, 3, 7 (all three boundary modes), 0.3-0.9 % at n = 4, 5, 6; 18: 0.7 %, 22 / 24: 1.6 / 5.5 % SLOWER.  (Round 4's single-placement sample of the old
// kernel had 20 rows 12 % slower: it crossed 168 registers then.)
#ifndef SG_ROLL_TILE_ROWS
#define SG_ROLL_TILE_ROWS 20
#endif
// the general one- and two-term forms of one output frame run as tiles too (derivative frames, orders 4-5 of small windows): 12-20 % faster
// than the walk at n = 2 ... 7 with one term, 12-20 % (n <= 5) / 1-5 % (n = 6, 7) with two (profiles/r04_2d_tile_experiments.txt)
#ifndef SG_ROLL_TILE_GENERAL
#define SG_ROLL_TILE_GENERAL 1
#endif
// additive form, half windows 8 .. 12 (A/B builds override).  End of round 5, after the x-stationary horizontal unit had freed 2N + 3 register pairs
// (tools/placement_2d.py, 32 frames of 4096^2, ms over six buffer pairs; profiles/r05_2d_tile_rows.txt): n = 8: 10 rows 0.900, 14 rows 0.862, 16 rows 0.889;
// n = 9: 10 / 12 / 14 rows 0.992 / 0.967 / 0.949; n = 10: 8 / 10 / 12 rows 1.073 / 1.017 / 1.016; n = 11, 12: 8-12-row tiles 1.38-1.62 against the walk's
// 1.27 / 1.30 at THREE waves per SIMD (67-127 spilled registers).  Two waves per SIMD with taller tiles at n <= 10 (round 6): n = 7 28 / 32 rows 1.68 against
// 1.585, n = 8 20 / 24 rows 1.95 / 1.81 against 1.64, n = 9 24 rows 1.85-1.87 against 1.82-1.86, n = 10 24 / 28 / 32 rows 1.95-1.96 against 1.99-2.02: left alone
#ifndef SG_ROLL_TR8
#define SG_ROLL_TR8 14
#endif
#ifndef SG_ROLL_TR9
#define SG_ROLL_TR9 14
#endif
#ifndef SG_ROLL_TR10
#define SG_ROLL_TR10 10     /* round 6: 12 rows spill 4 registers now that nothing else does; 10 rows were level (1.017 against 1.016 ms) */
#endif
// Half windows 11 .. 16 at TWO waves per SIMD (256 registers): the taller the tile, the fewer halo rows per output row, up to what the registers hold
// (4 (TR + 2N) + ~45; 20 rows spill at n = 16, 18 at n = 14).  64 frames of 4096^2, ms, tools/ab_2d.py (profiles/r06_2d_tiles_n11_16.txt):
//   n = 11: walk 2.47, 12 rows 2.28-2.35, 16: 2.13, 20: 2.08, 22 / 24: 2.10        n = 12: walk 2.55, 12: 2.38-2.43, 16: 2.24, 20: 2.17-2.19, 22: 2.20, 24: 2.17 (spills)
//   n = 13: walk 2.78, 8: 2.95, 12: 2.64, 14: 2.56, 16: 2.52, 18: 2.46             n = 14: walk 2.84, 12: 2.91, 14: 2.80, 16: 2.68, 18: 3.10 (spills)
//   n = 15: walk 3.12, 8: 2.97-3.03, 12: 2.72-2.78, 14: 2.74, 16: 2.61, 18: 2.52, 20: 2.60      n = 16: walk 3.11-3.18, 12: 2.93-2.97, 14: 2.80, 16: 2.88, 18: 2.74, 20: 2.80 (spills)
#ifndef SG_ROLL_TR11
#define SG_ROLL_TR11 20
#endif
#ifndef SG_ROLL_TR12
#define SG_ROLL_TR12 20
#endif
#ifndef SG_ROLL_TRG11
#define SG_ROLL_TRG11 0          /* A/B builds: one-term general form (derivative frames), tile rows for every half window 11 .. 16 (0 = the table in roll_tile_rows) */
#endif
// one-term general form (derivative frames), half windows 8 .. 10, d = (0,2), 64 frames, ms: three waves per SIMD with 12 / 10 / 10 rows (rounds 4-5) 1.67 / 1.88 / 1.95;
// TWO waves per SIMD: 16 rows 1.63 / 1.71 / 1.80, 20 rows 1.56 / 1.67 / 1.70-1.73, 24 rows 1.53 / 1.66 / 1.72, 28 rows 1.52 / 1.65 / 1.71
#ifndef SG_ROLL_TRG8
#define SG_ROLL_TRG8 24
#endif
#ifndef SG_ROLL_TRG9
#define SG_ROLL_TRG9 24
#endif
#ifndef SG_ROLL_TRG10
#define SG_ROLL_TRG10 20
#endif
#ifndef SG_ROLL_TRN2T2
#define SG_ROLL_TRN2T2 0         /* A/B builds: the fused two-output form with TWO terms per frame (gradient of order 3, 4), half windows <= 8: tile rows (0 = the table in roll_tile_rows) */
#endif
#ifndef SG_ROLL_TRN2
#define SG_ROLL_TRN2 0           /* A/B builds: the fused two-output form (gradient) with one term per frame, tile rows for every half window >= 8 (0 = the table in roll_tile_rows) */
#endif
#ifndef SG_ROLL_W2G_FROM
#define SG_ROLL_W2G_FROM 8       /* the one-term general form runs two waves per SIMD from this half window on */
#endif
#ifndef SG_ROLL_TR13
#define SG_ROLL_TR13 0           // A/B builds: tile rows for every half window 13 .. 16 (0 = the table in roll_tile_rows)
#endif
constexpr int roll_tile_rows(int n, int nt, int nout, bool box)
{
    if (box && nt == 2 && nout == 1 && n <= 7) return SG_ROLL_TILE_ROWS;
    // half window 8: 16 + 16 rows do not fit three waves per SIMD, 10 + 16 do (with 20 bytes of scratch): 6.92 vs 7.61 ms per 256 frames
    // (12 rows 6.93, 8 rows 7.21; profiles/r04_2d_tile_experiments.txt)
    if (box && nt == 2 && nout == 1 && n == 8) return SG_ROLL_TR8;
    if (SG_ROLL_TILE_GENERAL && !box && nout == 1 && nt == 1 && n == 8) return SG_ROLL_TRG8;
    // half windows 9, 10 (64 frames, ms, tile vs walk): n = 9 additive 1.97 vs 2.25, one term 1.85 vs 2.09 (10 rows; 8 rows 2.05 / 1.95);
    // n = 10 additive 2.14 vs 2.31 (8 rows; 10 spill: 2.16), one term 1.91 vs 2.13 (10 rows, 20 bytes of scratch; 8 rows 2.04)
    if (box && nt == 2 && nout == 1 && (n == 9 || n == 10)) return n == 9 ? SG_ROLL_TR9 : SG_ROLL_TR10;
    if (box && nt == 2 && nout == 1 && (n == 11 || n == 12)) return n == 11 ? SG_ROLL_TR11 : SG_ROLL_TR12;
    if (box && nt == 2 && nout == 1 && n >= 13) return SG_ROLL_TR13 ? SG_ROLL_TR13 : (n == 14 ? 16 : 18);
    if (SG_ROLL_TILE_GENERAL && !box && nout == 1 && nt == 1 && (n == 9 || n == 10)) return n == 9 ? SG_ROLL_TRG9 : SG_ROLL_TRG10;
    // one-term general form (derivative frames d = (0,1), (1,0), (0,2), ...) at two waves per SIMD, d = (0,2), 64 frames, ms (profiles/r06_2d_tiles_n11_16.txt):
    // n = 11: walk 2.03, 16 rows 1.84, 20 rows 1.76; 12: 2.05 / 1.91 / 1.85; 13: 2.16 / 2.03 / 1.97; 14: 2.26 / 2.12 / 2.07; 15: 2.30-2.32, 12 / 14 / 16 rows 2.46 / 2.33 / 2.30
    // (20 spill); 16: 2.31-2.36, 12 / 14 / 16 rows 2.57 / 2.41 / 2.46: the walk stays at 15 and 16
    if (SG_ROLL_TILE_GENERAL && !box && nout == 1 && nt == 1 && n >= 11) return SG_ROLL_TRG11 ? SG_ROLL_TRG11 : (n <= 14 ? 20 : 0);
    if (SG_ROLL_TILE_GENERAL && !box && nout == 1 && nt <= 2 && n <= 7) return SG_ROLL_TILE_ROWS;
    // the fused two- / three-output forms with one term per frame (gradient of order <= 2, Hessian of order <= 3): the three Hessian frames
    // of 64 x 4096^2 at n = 7 in 3.54 ms instead of 4.02 (2 waves per SIMD: 178 registers)
    if (SG_ROLL_TILE_GENERAL && !box && nout >= 2 && nt == 1 && n <= 7) return SG_ROLL_TILE_ROWS;
    // two terms per frame (the gradient of an order-3 or -4 filter -- the usual cubic), 64 frames, ms, walk / 16-row / 20-row tiles (tools/ab_2d_gradient.py --order 3; the
    // tile's frames are the walk's bits): n = 2: 2.26 / 2.10 / 2.41; 3: 2.71 / 2.13 / 2.17; 5: 3.08 / 2.64 / 2.79; 7: 3.63 / 3.45 / 3.46; 8: 4.20 / 4.17 / 3.87
    if (SG_ROLL_TILE_GENERAL && !box && nout == 2 && nt == 2 && n <= 8) return SG_ROLL_TRN2T2 ? SG_ROLL_TRN2T2 : (n == 8 ? 20 : 16);
    // the same form at half windows 8 .. 12 (gradient of order <= 2, 64 frames, ms, tools/ab_2d_gradient.py; the tile's frames are the walk's bits):
    // n = 8: walk 2.71, 16 rows 2.51, 20 rows 2.63; 9: 3.16 / 2.73 / 2.75; 10: 3.11 / 2.91 / 2.93; 11: 3.54 / 3.19 / 3.22; 12: 3.51 / 3.21 / 3.42 (spills)
    if (SG_ROLL_TILE_GENERAL && !box && nout == 2 && nt == 1 && n >= 8) return SG_ROLL_TRN2 ? SG_ROLL_TRN2 : (n <= 12 ? 16 : 0);
    return 0;
}
#ifndef SG_ROLL_TILE_WAVES
#define SG_ROLL_TILE_WAVES 3
#endif
// (the general two-term form at n = 6, 7 spills 36-52 bytes at 3 waves per SIMD)
#ifndef SG_ROLL_W2_FROM
#define SG_ROLL_W2_FROM 11       /* tiles of half windows >= this run at TWO waves per SIMD (256 registers: taller tiles) */
#endif
constexpr int roll_tile_waves(int n, int nt = 2, bool box = true, int nout = 1)
{
    return (nout >= 2 || (!box && nt == 2 && n >= 6) || (!box && nt == 1 && n >= SG_ROLL_W2G_FROM) || n >= SG_ROLL_W2_FROM) ? 2 : SG_ROLL_TILE_WAVES;
}

// waves per block: the waves of a block walk neighbouring strips row for row, so a block's loads of one row step are one
// contiguous run of the frame row
#ifndef SG_ROLL_WPB
#define SG_ROLL_WPB 4
#endif
// (tile form: 2 -- a block retires when its slowest wave does, and 2 of a frame row's 18 strips are edge strips; 64 / 256 frames of
//  4096^2 at n = 7: 1 wave 1.65 / 6.85 ms, 2 waves 1.61 / 6.58, 4 waves 1.69 / 6.84, 8 waves 2.05 / 7.85)
#ifndef SG_ROLL_TILE_WPB
#define SG_ROLL_TILE_WPB 2
#endif
constexpr int roll_wpb(int n, int tr = 0) { (void)n; return tr > 0 ? SG_ROLL_TILE_WPB : SG_ROLL_WPB; }

template <int N, int NT, int NOUT, bool BOX, bool ACC, int TR = 0>
__global__ __launch_bounds__(64 * roll_wpb(N, TR), TR > 0 ? roll_tile_waves(N, NT, BOX, NOUT) : roll_min_waves(N, NT, NOUT)) void sg2d_rolling_kernel(const Job2D job, const RollTaps<N, NT, NOUT> taps, float *const out1, float *const out2,
                                                           unsigned strips, unsigned bands, int band_rows, unsigned total_items, int aligned)
{
    typedef Roll<N> R;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *mine = lds + wv * (2 * NOUT * NT * R::BUFW);      // two LDS rows per output and term, private to this wave

    // one item per wave by default (the loop runs once), or persistent waves striding over the items; blocks that share an XCD
    // (blockIdx % 8) take neighbouring items (halo columns meet in L2)
    const unsigned nblk = gridDim.x;
    // bit 3: blocks in launch order (A/B: SAVGOL_HIP_ROLL_XCD=0).  Bits 8..: blocks per XCD CHUNK -- chunks are dealt to the XCDs round robin, so the eight
    // fronts stay within eight chunks of each other; 0 = every XCD sweeps one contiguous eighth of the launch (rounds 3-5)
    unsigned blk = blockIdx.x;
    if (!(aligned & 8)) {
        const unsigned chunk = (unsigned)aligned >> 8;
        if (chunk == 0) blk = (blk & 7u) * (nblk >> 3) + (blk >> 3);
        else {
            const unsigned span = 8u * chunk, q = blk / span;
            if ((q + 1u) * span <= nblk) { const unsigned r = blk - q * span; blk = (q * 8u + (r & 7u)) * chunk + (r >> 3); }    // the last, partial span keeps launch order
        }
    }
    constexpr unsigned WPB = (unsigned)roll_wpb(N, TR);
    const unsigned nwaves = nblk * WPB;

    const bool valid = job.boundary == SAVGOL2D_BOUNDARY_VALID;
    // the job's own half windows: a rectangular window runs on the square kernel of its larger half width with zero-padded factors
    const int xlo = valid ? job.nx : 0, xhi = valid ? job.cols - job.nx : job.cols;
    const int ylo = valid ? job.ny : 0, yhi = valid ? job.rows - job.ny : job.rows;

    // (tile form: one tile per wave, always -- with no loop nothing of the dispatch stays live across the tile's 170 registers)
    for (unsigned item = blk * WPB + (unsigned)wv; item < total_items; item += (TR > 0 ? 0xffffffffu - item : nwaves)) {
        const unsigned strip = item % strips, ib = item / strips;
        const unsigned band = ib % bands, img = ib / bands;
#ifdef SG_ROLL_SKIP_EDGE_STRIPS                                   // timing experiment only (wrong frames): what do the frame-edge strips cost?
        if (strip == 0 || strip + 1 == strips) continue;
#endif
        const int yb = (int)band * band_rows;
        const int nout = job.rows - yb < band_rows ? job.rows - yb : band_rows;
        const float *in = job.in + (long long)img * job.in_pitch;
        float *outs[NOUT];
        outs[0] = job.out + (long long)img * job.out_pitch;
        if constexpr (NOUT > 1) outs[1] = out1 + (long long)img * job.out_pitch;
        if constexpr (NOUT > 2) outs[2] = out2 + (long long)img * job.out_pitch;
        const int sx = (int)strip * R::SW;
        // fast variant: all 256 input columns inside the frame, all SW output columns stored, 16-byte aligned rows
        if ((aligned & 3) == 3 && sx - 4 * R::HL >= 0 && sx - 4 * R::HL + 256 <= job.cols && sx >= xlo && sx + R::SW <= xhi)
            roll_item<N, NT, NOUT, 1, BOX, ACC, TR>(job, taps, mine, in, outs, sx - 4 * R::HL, yb, nout, lane, xlo, xhi, ylo, yhi);
        else if constexpr (TR > 0) {
            // tile kernels are only launched on frames whose every strip can run on vector loads (aligned bits 0-2: 16-byte aligned rows, cols % 4 == 0,
            // wide enough for one reflection; launch_roll_kernel sends every other frame to the strip walk).  Round 6: the scalar path used to be
            // compiled into the tile kernels too, and IT was what spilled -- 2 to 24 registers, a private segment for every wave of a launch that
            // never ran it.
            if (valid) roll_item<N, NT, NOUT, 3, BOX, ACC, TR>(job, taps, mine, in, outs, sx - 4 * R::HL, yb, nout, lane, xlo, xhi, ylo, yhi);
            else roll_item<N, NT, NOUT, 2, BOX, ACC, TR>(job, taps, mine, in, outs, sx - 4 * R::HL, yb, nout, lane, xlo, xhi, ylo, yhi);
        } else
            roll_item<N, NT, NOUT, 0, BOX, ACC, TR>(job, taps, mine, in, outs, sx - 4 * R::HL, yb, nout, lane, xlo, xhi, ylo, yhi);
    }
}

// ---- host ----
// fill output o's taps from its factor block; false when a vector has no definite parity or the terms disagree
template <int N, int NT, int NOUT>
static bool fill_taps(RollTaps<N, NT, NOUT> &taps, int o, const float *factors, float scale)
{
    float sy = 0.0f, sx = 0.0f;
    for (int t = 0; t < NT; ++t) {
        const float *q = factors + (size_t)t * 2 * (2 * N + 2), *g = q + (2 * N + 2);
        float s1, s2;
        if (!vector_parity(g, N, &s1) || !vector_parity(q, N, &s2)) return false;
        if (t > 0 && (s1 != sy || s2 != sx)) return false;
        sy = s1; sx = s2;
        for (int k = 0; k <= N; ++k) {
            const float gk = (k == N && sy < 0.0f) ? 0.0f : g[k];
            const float qk = (k == N && sx < 0.0f) ? 0.0f : (float)((double)q[k] * (double)scale);
            if (k & 1) { taps.g[o][t][k >> 1].y = gk; taps.q[o][t][k >> 1].y = qk; }
            else       { taps.g[o][t][k >> 1].x = gk; taps.q[o][t][k >> 1].x = qk; }
        }
    }
    taps.sy[o] = f32x2{sy, sy};
    taps.sx[o] = f32x2{sx, sx};
    return true;
}

// Additive kernels (see roll_item, BOX): two terms, the first with a constant column factor G_0 and the second with a constant row
// factor Q_1 -- what sg2d_factors_from_kernel returns for W(x, y) = A(x) + B(y): its G_0 is the normalised constant and, A being
// orthogonal to G_1, Q_1 = sum_y G_1 B is the same for every x.  Rewritten as A'(x) + B'(y) with B'(0) = 0:
//      A'(x) = scale (g0 Q_0(x) + G_1(0) q1),   B'(y) = scale q1 (G_1(y) - G_1(0)).
template <int N>
static bool fill_box_taps(RollTaps<N, 2, 1> &taps, const float *factors, float scale)
{
    const float *q0 = factors, *g0 = q0 + (2 * N + 2), *q1 = g0 + (2 * N + 2), *g1 = q1 + (2 * N + 2);
    auto constant = [](const float *v) {
        float lo = v[0], hi = v[0];
        for (int k = 1; k <= 2 * N; ++k) { lo = fminf(lo, v[k]); hi = fmaxf(hi, v[k]); }
        return hi - lo <= 4e-6f * fmaxf(fabsf(lo), fabsf(hi)) && lo != 0.0f;
    };
    float s1, s2;
    if (!constant(g0) || !constant(q1) || !vector_parity(q0, N, &s1) || !vector_parity(g1, N, &s2) || s1 != 1.0f || s2 != 1.0f) return false;
    double g0c = 0.0, q1c = 0.0;
    for (int k = 0; k <= 2 * N; ++k) { g0c += g0[k]; q1c += q1[k]; }
    g0c /= 2 * N + 1; q1c /= 2 * N + 1;
    {   // Smoothing kernels only.  The summed Laplacian kernel of order <= 3 is additive too, but its taps cancel (sum 0): the box
        // sums are then large against the output and their rounding shows -- 5e-6 of the output on the test frames, the general
        // two-term form 2e-6.  Condition: |sum W| >= half of sum |W|.
        double sum = 0.0, sum_abs = 0.0;
        for (int y = 0; y <= 2 * N; ++y)
            for (int x = 0; x <= 2 * N; ++x) {
                const double w = g0c * (double)q0[x] + (double)g1[y] * q1c;
                sum += w; sum_abs += std::fabs(w);
            }
        if (std::fabs(sum) < 0.5 * sum_abs) return false;
    }
    for (int k = 0; k <= N; ++k) {
        // symmetric vectors: average the two halves (they differ in the last bit of the float factors)
        const double qa = 0.5 * ((double)q0[k] + (double)q0[2 * N - k]), ga = 0.5 * ((double)g1[k] + (double)g1[2 * N - k]);
        const float a = (float)((double)scale * (g0c * qa + (double)g1[N] * q1c));
        const float b = (float)((double)scale * q1c * (ga - (double)g1[N]));
        if (k & 1) { taps.q[0][0][k >> 1].y = a; taps.g[0][1][k >> 1].y = b; }
        else       { taps.q[0][0][k >> 1].x = a; taps.g[0][1][k >> 1].x = b; }
        // the x-stationary table: the same float on both sides of the centre
        taps.qx[k].x = a; taps.qx[2 * N - k].x = a;
        taps.qx[k + 1].y = a; taps.qx[2 * N - k + 1].y = a;
    }
    taps.qx[0].y = 0.0f; taps.qx[2 * N + 1].x = 0.0f;
    taps.sy[0] = f32x2{1.0f, 1.0f};
    taps.sx[0] = f32x2{1.0f, 1.0f};
    return true;
}

template <int N, int NT, int NOUT, bool BOX, bool ACC = false, int TR = 0>
static int launch_roll_kernel(const Job2D &job, const RollTaps<N, NT, NOUT> &taps, float *out1, float *out2, unsigned images, int cu_count, hipStream_t st)
{
    typedef Roll<N> R;
    int aligned = 0;
    if (job.in_stride % 4 == 0 && job.in_pitch % 4 == 0 && (reinterpret_cast<uintptr_t>(job.in) & 15u) == 0) aligned |= 1;
    if (job.out_stride % 4 == 0 && job.out_pitch % 4 == 0 && (reinterpret_cast<uintptr_t>(job.out) & 15u) == 0 &&
        (NOUT < 2 || (reinterpret_cast<uintptr_t>(out1) & 15u) == 0) && (NOUT < 3 || (reinterpret_cast<uintptr_t>(out2) & 15u) == 0) &&
        (long long)job.rows * job.out_stride * 4 < 0x7fffff00ll) aligned |= 2;       // the store descriptor holds a 31-bit byte count
    if (TR > 0 && job.cols % 4 == 0 && job.cols >= 32) aligned |= 4;
    if constexpr (TR > 0) {
        // frames a tile kernel cannot take on vector loads alone (odd strides, unaligned bases, cols % 4 != 0, narrower than 32 columns): the strip walk
        if ((aligned & 7) != 7) return launch_roll_kernel<N, NT, NOUT, BOX, ACC, 0>(job, taps, out1, out2, images, cu_count, st);
    }
    const unsigned strips = (unsigned)((job.cols + R::SW - 1) / R::SW);
    static int per_cu = 0;                                   // resident blocks per CU of this instantiation
    constexpr unsigned WPB = (unsigned)roll_wpb(N, TR);
    size_t lds = sizeof(float) * WPB * 2 * NOUT * NT * R::BUFW;
    if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sg2d_rolling_kernel<N, NT, NOUT, BOX, ACC, TR>, 64 * WPB, lds) != hipSuccess || nb < 1)
            nb = WPB <= 8 ? 2 : 1;
        per_cu = nb > (int)(16 / WPB) ? (int)(16 / WPB) : nb;
    }
    const unsigned nwaves = (unsigned)cu_count * (unsigned)per_cu * WPB;
    // One item per wave, blocks handed out by the hardware dispatcher in order -- as in the 1-D kernel, and for the same reason: the
    // same items and bands on a persistent grid (resident waves striding over the items) were 1-5 % slower at every half window
    // (n = 4: 1.89 vs 1.79 ms per 64 frames; round 3).  The band count is still chosen for whole rounds of the resident waves, which is
    // also what keeps the tail of the dispatch short.
    // a launch indexes < 2^32 threads (HIP rejects gridDim.x * blockDim.x >= 2^32) and < 2^32 items: split over images
    unsigned long long per_image;
    unsigned bands;
    int band_rows;
    auto geometry = [&](unsigned long long imgs) {
        bands = choose_bands(job.rows, imgs * strips, nwaves, N, 0.3);       // warm-up rows are only loaded
        band_rows = (int)((job.rows + (int)bands - 1) / (int)bands);
        // the additive form re-seeds its rolling column sums every U rows of a band: bands that start on multiples of U keep that
        // phase tied to the FRAME row, so a frame's bits do not depend on how many frames share the launch (the band count does)
        if (BOX) band_rows = (band_rows + R::U - 1) / R::U * R::U;
        if (TR > 0) band_rows = TR;                          // tiles: TR rows each, whatever the batch (the re-seed phase is the tile's first row)
        bands = (unsigned)((job.rows + band_rows - 1) / band_rows);
        per_image = (unsigned long long)strips * bands;
    };
    geometry(images);
    const unsigned long long max_items = ((1ull << 32) - 4096) / 64;           // items == waves when not persistent
    unsigned long long img_step = per_image ? max_items / per_image : images;
    if (img_step == 0) { sg_set_error("2-D frame too large for one launch (%llu items)", per_image); return -1; }
    if (img_step < images) geometry(img_step); else img_step = images;
    for (unsigned long long i0 = 0; i0 < images; i0 += img_step) {
        const unsigned long long ni = images - i0 < img_step ? images - i0 : img_step;
        const unsigned long long total = ni * per_image;
        unsigned grid = (unsigned)((total + WPB - 1) / WPB);
        grid = (grid + 7u) & ~7u;
        Job2D part = job;
        part.in = job.in + (long long)i0 * job.in_pitch;
        part.out = job.out + (long long)i0 * job.out_pitch;
        // XCD chunk (round 5): whole frames, at least 128 bands of every strip, dealt to the XCDs round robin -- the eight fronts stay within a few frames of
        // each other instead of an eighth of the stack apart, and no halo row crosses a chunk.  Sampled over fresh buffer pairs in one process, columns
        // rotated (tools/placement_2d.py, profiles/r05_placement_2d.txt; 64 frames of 4096^2, n = 7): 1.624 against 1.632 ms median for contiguous eighths
        // (first pass over a fresh pair 1.641 / 1.669) -- level to slightly ahead; 64-band chunks 1.645.
        int aligned_launch = aligned;
        const int chunk_bands = (int)(bands * ((128u + bands - 1u) / bands));
        if (chunk_bands > 0 && TR > 0) {
            const unsigned long long chunk_items = (unsigned long long)chunk_bands * strips;
            if (chunk_items % WPB == 0 && chunk_items / WPB < (1u << 22) && chunk_items / WPB * 8u <= grid) aligned_launch |= (int)((unsigned)(chunk_items / WPB) << 8);
        }
        hipLaunchKernelGGL((sg2d_rolling_kernel<N, NT, NOUT, BOX, ACC, TR>), dim3(grid), dim3(64 * WPB), lds, st, part, taps,
                           out1 ? out1 + (long long)i0 * job.out_pitch : nullptr, out2 ? out2 + (long long)i0 * job.out_pitch : nullptr, strips, bands,
                           band_rows, (unsigned)total, aligned_launch);
    }
    return 0;
}

template <int N, int NT, int NOUT>
static int launch_roll(const Job2D &job, const float *const (&factors)[NOUT], const float (&scale)[NOUT], float *out1, float *out2, unsigned images,
                       int cu_count, hipStream_t st)
{
    if (job.accumulate) {
        // the second pass of a kernel split over two launches (sg_2d.hip, roll_passes): built where a split can be asked for
        if constexpr (NOUT == 1 && N >= 9 && NT <= 2) {
            RollTaps<N, NT, 1> taps;
            memset(&taps, 0, sizeof(taps));
            if (!fill_taps<N, NT, 1>(taps, 0, factors[0], scale[0])) return 1;
            return launch_roll_kernel<N, NT, 1, false, true>(job, taps, out1, out2, images, cu_count, st);
        } else return 1;
    }
    if constexpr (NT == 2 && NOUT == 1) {
        RollTaps<N, 2, 1> box;
        memset(&box, 0, sizeof(box));
        if (fill_box_taps<N>(box, factors[0], scale[0])) {
            constexpr int TR = roll_tile_rows(N, 2, 1, true);
            if constexpr (TR > 0) {
                // SAVGOL_HIP_ROLL_TILE=0: the strip walk of rounds 1-3 (A/B runs)
                static const int tile_env = [] { const char *e = getenv("SAVGOL_HIP_ROLL_TILE"); return e ? atoi(e) : 1; }();
                if (tile_env != 0) return launch_roll_kernel<N, 2, 1, true, false, TR>(job, box, out1, out2, images, cu_count, st);
            }
            return launch_roll_kernel<N, 2, 1, true>(job, box, out1, out2, images, cu_count, st);
        }
    }
    RollTaps<N, NT, NOUT> taps;
    memset(&taps, 0, sizeof(taps));
    for (int o = 0; o < NOUT; ++o)
        if (!fill_taps<N, NT, NOUT>(taps, o, factors[o], scale[o])) return 1;
    constexpr int TRG = roll_tile_rows(N, NT, NOUT, false);
    if constexpr (TRG > 0) {
        static const int tile_env = [] { const char *e = getenv("SAVGOL_HIP_ROLL_TILE"); return e ? atoi(e) : 1; }();
        if (tile_env != 0) return launch_roll_kernel<N, NT, NOUT, false, false, TRG>(job, taps, out1, out2, images, cu_count, st);
    }
    return launch_roll_kernel<N, NT, NOUT, false>(job, taps, out1, out2, images, cu_count, st);
}

template <int N, int NT>
static int dispatch_roll(int n, int terms, const Job2D &job, const float *factors, float scale, unsigned images, int cu_count, hipStream_t st)
{
    if (n == N && terms == NT) {
        const float *const f1[1] = {factors};
        const float s1[1] = {scale};
        return launch_roll<N, NT, 1>(job, f1, s1, nullptr, nullptr, images, cu_count, st);
    }
    if constexpr (NT < roll_max_terms(N, 1)) return dispatch_roll<N, NT + 1>(n, terms, job, factors, scale, images, cu_count, st);
    else if constexpr (N < SEP_ROLL_MAX_N) return dispatch_roll<N + 1, 1>(n, terms, job, factors, scale, images, cu_count, st);
    else return 1;
}

// two output frames from one walk over the input, the same number of terms for both: the gradient
template <int N, int NT>
static int dispatch_roll2(int n, int terms, const Job2D &job, const float *f0, float s0, const float *f1, float s1, float *out1, unsigned images,
                          int cu_count, hipStream_t st)
{
    if (n == N && terms == NT) {
        const float *const ff[2] = {f0, f1};
        const float ss[2] = {s0, s1};
        return launch_roll<N, NT, 2>(job, ff, ss, out1, nullptr, images, cu_count, st);
    }
    if constexpr (NT < roll_max_terms(N, 2)) return dispatch_roll2<N, NT + 1>(n, terms, job, f0, s0, f1, s1, out1, images, cu_count, st);
    else if constexpr (N < SEP_ROLL_MAX_N) return dispatch_roll2<N + 1, 1>(n, terms, job, f0, s0, f1, s1, out1, images, cu_count, st);
    else return 1;
}

// 0 = launched, 1 = this object does not cover the case (half window outside SEP_ROLL_MIN_N..SEP_ROLL_MAX_N, no
// definite parity).  Built once per half-window group (Makefile), each under its own name SEP_ROLL_FN.
int SEP_ROLL_FN(int n, int terms, const Job2D &job, const float *factors, float scale, unsigned images, int cu_count, hipStream_t st)
{
    if (n < SEP_ROLL_MIN_N || n > SEP_ROLL_MAX_N || terms < 1 || terms > roll_max_terms(n, 1)) return 1;
    return dispatch_roll<SEP_ROLL_MIN_N, 1>(n, terms, job, factors, scale, images, cu_count, st);
}

// the two-output form: job.out and out1 share stride and pitch; both outputs have `terms` terms
int SEP_ROLL_FN2(int n, int terms, const Job2D &job, const float *factors0, float scale0, const float *factors1, float scale1, float *out1,
                 unsigned images, int cu_count, hipStream_t st)
{
    if (n < SEP_ROLL_MIN_N || n > SEP_ROLL_MAX_N || terms < 1 || terms > roll_max_terms(n, 2)) return 1;
    return dispatch_roll2<SEP_ROLL_MIN_N, 1>(n, terms, job, factors0, scale0, factors1, scale1, out1, images, cu_count, st);
}

#ifdef SG_STAMPS2D
}  // namespace sg
extern "C" __attribute__((visibility("default"))) int savgol_hip_debug_set_stamps2d(void *d_buffer_512_u64)
{
    unsigned long long *p = static_cast<unsigned long long *>(d_buffer_512_u64);
    return hipMemcpyToSymbol(HIP_SYMBOL(sg::g_stamps2d), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
namespace sg {
#endif

}  // namespace sg
