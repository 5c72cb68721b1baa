// sg_k1d_moment.hpp -- the fp32 inner product for half windows 24..32 with ~42 multiply-adds per output instead of 2n+1
// (43 instead of 65 at n = 32, poly_order 4).
//
// Why.  At these windows sg1d_center_kernel is not limited by HBM but by the energy of its FMAs: at n = 32 the vector unit is
// issue-saturated (4.25 cycles per VALU instruction per SIMD, PMC) at the 1.5 GHz the chip holds under this load
// (profiles/r02_*).  Removing index arithmetic did not help (-10 % instructions, -0.6 % time): only fewer
// FMAs do.
//
// How.  Savitzky-Golay centre taps are samples of a polynomial of degree <= poly_order in the tap index k
// (reference compute_weight, src/savgolFilter.c:336-356: a sum of Gram polynomials F_j(k)).  A lane owns 32 consecutive
// outputs r = 0..31; its window is samples X[0 .. 32 + 2n + OFF) and output r reads X[r + OFF + k] with tap k.  The samples
// X[LO .. HI) (LO = 31 + OFF rounded up to even, HI = OFF + 2n + 1 rounded down to even: 32..64 at n = 32, 32..56 at n = 28,
// 32..48 at n = 24) lie inside the window of EVERY one of the lane's outputs, and on that common block the taps
// w[LO + t - r - OFF] are a polynomial q_r(t) of degree < M1.  Written in a basis phi_s(t) (Legendre polynomials on the block),
//     sum_t q_r(t) X[LO+t]  =  sum_{s<M1} c_s(r) * mu_s,      mu_s = sum_t phi_s(t) X[LO+t]     (block moments, once per lane)
// so the HI-LO taps that fall on the block cost M1 multiply-adds per output plus M1 per block sample for the moments, and only
// the 31 or 33 taps outside it are applied one by one: 33 + 5 + 5 = 43 at n = 32, poly_order 4, instead of 65.
// The moments are LOCAL (one block of at most 32 samples, no recurrence, no running sums), which is why fp32 holds: measured
// normwise error vs the fp64 oracle 3.0e-7 ... 4.5e-7 for smoothing filters, the same as the plain sum (tests/test_gpu_1d.py).
//
// The host (sg1d_moment_prepare, sg_k1d_moment_fit.cpp) fits the polynomial to the filter's fp32 table in double, refuses tables
// that are not a polynomial to 3e-7 of max|w| (hand-edited tables run the plain kernel), and uploads
//     w[66] | phi[s-1][t], s = 1..6, t < (HI-LO)/2 | c[s][J] = (c_s(2J), c_s(2J+1)), s = 0..6, J = 0..15
// (layout in sg_k1d_host.hpp).  Everything is read through scalar loads and lives in SGPRs.
#pragma once

#include "sg_k1d.hpp"

namespace sg {

typedef f32x2 __attribute__((address_space(4))) ConstPairM;

// both halves of the constant pair are used: acc.lo += c.lo * x.lo, acc.hi += c.hi * x.hi
__device__ __forceinline__ void pk_fma_pair(f32x2 &acc, const f32x2 c, const f32x2 x)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "s"(c), "v"(x));
}
__device__ __forceinline__ f32x2 pk_mul_pair(const f32x2 c, const f32x2 x)
{
    f32x2 p;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "s"(c), "v"(x));
    return p;
}
// the mirrored pair: acc.lo += SIGN * c.hi * x.lo, acc.hi += SIGN * c.lo * x.hi
template <bool NEG>
__device__ __forceinline__ void pk_fma_pair_swapped(f32x2 &acc, const f32x2 c, const f32x2 x)
{
    if constexpr (NEG) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(acc) : "s"(c), "v"(x));
    else               asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(c), "v"(x));
}
// acc += c * x.lo (x.lo broadcast to both halves)
__device__ __forceinline__ void pk_fma_pair_bcast(f32x2 &acc, const f32x2 c, const f32x2 x)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(c), "v"(x));
}

template <int N, int M1>
struct MomentConv {
    typedef K1D<float, N> K;
    typedef MomentArgs Args;
    static constexpr int OFF = K::OFF, LO = moment_lo(N), HI = moment_hi(N), BK = HI - LO, NPB = BK / 2;   // NPB pairs in the block
    static constexpr int WS = 2 * N + 1;
    // partial sums per output (round 4): the taps applied one by one go to chain (tap index) mod 3, the block's moment terms to
    // chain (moment index) mod 3, joined once at the end -- Conv<float> in sg_k1d.hpp says why three interleaved chains
    static constexpr int CH = 3;
    static_assert(N >= MOMENT_MIN_N && N <= MOMENT_MAX_N && K::R == 32 && K::VPL == 8, "32 outputs per lane, 8 vectors per lane");
    static_assert(OFF == moment_off(N) && LO % 2 == 0 && HI % 2 == 0 && BK % 4 == 0 && BK >= 16 && BK <= 32, "block geometry");
    static_assert(LO >= 31 + OFF && HI <= OFF + 2 * N + 1, "the block must lie inside every output's window");
    static_assert(M1 >= 1 && M1 <= MOMENT_MAX_TERMS, "1..7 moments");
    // taps the head (samples below LO) and the tail (samples from HI on) use, as ranges of SGPR pairs of the table
    static constexpr int HEAD_PAIRS = (LO - 1 - OFF) / 2 + 1;                    // taps 0 .. LO-1-OFF
    static constexpr int TAIL_K0 = (HI - 1 - 30 - OFF) < 0 ? 0 : (HI - 1 - 30 - OFF);   // smallest tap the tail touches (output 30/31)
    static constexpr int TAIL_P0 = TAIL_K0 / 2, TAIL_PAIRS = N - TAIL_P0 + 1;    // pairs TAIL_P0 .. N (tap 2N sits in pair N)

    // `count` aligned pairs from the table, pinned to this place in the instruction stream: the empty asm ties the address
    // to a value computed just before (hoisted to the top of the kernel, the ~290 scalar registers of constants would spill)
    template <int COUNT>
    static __device__ __forceinline__ void load_pairs(f32x2 (&dst)[COUNT], const float *p, const f32x2 after)
    {
        asm volatile("" : "+s"(p) : "v"(after));
        const ConstPairM *cp = reinterpret_cast<const ConstPairM *>(reinterpret_cast<uintptr_t>(p));
#pragma unroll
        for (int i = 0; i < COUNT; ++i) dst[i] = cp[i];
    }

    // the pair (X[I], X[I+1]) reaches accumulator pair J with tap k = I - 2J - OFF (even output from X[I], odd from X[I+1])
    template <int I, int P0, int NPAIRS, int J = 0>
    static __device__ __forceinline__ void feed(f32x2 (&A)[CH][16], const f32x2 (&W)[NPAIRS], const f32x2 x)
    {
        if constexpr (J < 16) {
            constexpr int k = I - 2 * J - OFF;
            if constexpr (k >= 0 && k <= 2 * N) {
                static_assert((k >> 1) >= P0 && (k >> 1) - P0 < NPAIRS, "tap outside the loaded range");
                // taps 0 .. CH-1 open their chains (only the head feeds them, and only as whole pairs: needs_zero() below)
                static_assert(k >= CH || P0 == 0, "the tail never sees taps 0 .. CH-1");
                if constexpr (k < CH) A[k][J] = pk_mul_sgpr<(k & 1)>(W[(k >> 1) - P0], x);
                else pk_fma_sgpr<(k & 1)>(A[k % CH][J], W[(k >> 1) - P0], x);
            }
            feed<I, P0, NPAIRS, J + 1>(A, W, x);
        }
    }
    // one sample reaches only one half of every pair: X[I] for the even outputs (HALF 0), X[I+1] for the odd ones (HALF 1)
    template <int I, int HALF, int P0, int NPAIRS>
    static __device__ __forceinline__ void feed_single(f32x2 (&A)[CH][16], const f32x2 (&W)[NPAIRS], const float x)
    {
        static_for<16>([&](auto jc) -> bool {
            constexpr int J = decltype(jc)::value;
            constexpr int k = I - 2 * J - OFF;
            if constexpr (k >= 0 && k <= 2 * N) {
                static_assert((k >> 1) >= P0 && (k >> 1) - P0 < NPAIRS, "tap outside the loaded range");
                const float w = (k & 1) ? W[(k >> 1) - P0].y : W[(k >> 1) - P0].x;
                if constexpr (HALF == 0) A[k % CH][J].x = __builtin_fmaf(w, x, A[k % CH][J].x);
                else                     A[k % CH][J].y = __builtin_fmaf(w, x, A[k % CH][J].y);
            }
            return true;
        });
    }

    static __device__ __forceinline__ float4 vec(const char *win, int q) { return *reinterpret_cast<const float4 *>(win + slab_vec_off<8>(q)); }

    static __device__ __forceinline__ void run(const char *win, const MomentArgs &args, float (&acc)[32], unsigned)
    {
        const float *tab = args.table;
        // chain c of pair J is opened by the head's whole-pair feed of tap c, sample pair (2J + c + OFF, +1) -- unless that pair reaches the
        // block (the last one or two pairs): those chains start from zero
        f32x2 A[CH][16];
        static_for<CH * 16>([&](auto ic) -> bool {
            constexpr int c = decltype(ic)::value / 16, J = decltype(ic)::value % 16;
            if constexpr (2 * J + c + OFF + 1 >= LO) A[c][J] = f32x2{0.0f, 0.0f};
            return true;
        });

        // ---- 1. head: samples below LO ----
        {
            f32x2 W[HEAD_PAIRS];
            load_pairs<HEAD_PAIRS>(W, tab + MOMENT_OFF_W, f32x2{0.0f, 0.0f});
            f32x2 prev = f32x2{0.0f, 0.0f};
            static_for<(LO + 3) / 4>([&](auto qc) -> bool {
                constexpr int q = decltype(qc)::value;
                const float4 v = vec(win, q);
                const f32x2 e0 = {v.x, v.y}, e1 = {v.z, v.w};
                // pair positions of this vector: I = 4q-1 (straddle with the previous one), 4q, 4q+1 (straddle), 4q+2
                if constexpr (q > 0 && 4 * q < LO) feed<4 * q - 1, 0, HEAD_PAIRS>(A, W, pk_straddle(prev, e0));
                if constexpr (q > 0 && 4 * q == LO) feed_single<4 * q - 1, 0, 0, HEAD_PAIRS>(A, W, prev.y);     // X[LO-1] pairs with a block sample
                if constexpr (4 * q + 1 < LO) feed<4 * q, 0, HEAD_PAIRS>(A, W, e0);
                if constexpr (4 * q + 2 < LO) feed<4 * q + 1, 0, HEAD_PAIRS>(A, W, pk_straddle(e0, e1));
                if constexpr (4 * q + 2 == LO) feed_single<4 * q + 1, 0, 0, HEAD_PAIRS>(A, W, v.y);
                if constexpr (4 * q + 3 < LO) feed<4 * q + 2, 0, HEAD_PAIRS>(A, W, e1);
                prev = e1;
                return true;
            });
            // LO a multiple of 4 and the loop above stopped before the vector that starts at LO: X[LO-1] is prev.y
            if constexpr (LO % 4 == 0) feed_single<LO - 1, 0, 0, HEAD_PAIRS>(A, W, prev.y);
        }

        // ---- 2. tail: samples from HI on ----
        {
            f32x2 W[TAIL_PAIRS];
            load_pairs<TAIL_PAIRS>(W, tab + MOMENT_OFF_W + 2 * TAIL_P0, A[0][0]);
            f32x2 prev = f32x2{0.0f, 0.0f};
            constexpr int Q0 = (HI - 1) / 4;                                // the vector that holds X[HI-1]
            static_for<K::WQ - Q0>([&](auto qc) -> bool {
                constexpr int q = Q0 + decltype(qc)::value;
                const float4 v = vec(win, q);
                const f32x2 e0 = {v.x, v.y}, e1 = {v.z, v.w};
                if constexpr (4 * q == HI) feed_single<4 * q - 1, 1, TAIL_P0, TAIL_PAIRS>(A, W, v.x);       // X[HI] pairs with a block sample
                if constexpr (4 * q - 1 >= HI && q > Q0) feed<4 * q - 1, TAIL_P0, TAIL_PAIRS>(A, W, pk_straddle(prev, e0));
                if constexpr (4 * q >= HI) feed<4 * q, TAIL_P0, TAIL_PAIRS>(A, W, e0);
                if constexpr (4 * q + 2 == HI) feed_single<4 * q + 1, 1, TAIL_P0, TAIL_PAIRS>(A, W, v.z);
                if constexpr (4 * q + 1 >= HI) feed<4 * q + 1, TAIL_P0, TAIL_PAIRS>(A, W, pk_straddle(e0, e1));
                if constexpr (4 * q + 2 >= HI) feed<4 * q + 2, TAIL_P0, TAIL_PAIRS>(A, W, e1);
                prev = e1;
                return true;
            });
        }
        // ---- 3. the taps applied one by one are done: join their three chains.  The block's share comes LAST and through its own short chain
        //         (smallest moment first): it is the largest single term of an output that cancels -- the moving average that nulls a tone,
        //         a derivative -- and every add made AFTER it rounds at its magnitude.  Added early, as rounds 2-3 did, the moment kernels
        //         were up to 2.8 x the reference's own error where the plain kernel is at 1.0 (tools/diag_1d_accuracy.py). ----
#pragma unroll
        for (int j = 0; j < 16; ++j) A[0][j] = (A[0][j] + A[1][j]) + A[2][j];

        // ---- 4. moments of the block X[LO..HI): M[s] = (sum over even t, sum over odd t) of phi_s(t) X[LO+t] ----
        f32x2 M[M1];
        {
            constexpr int NP = M1 > 1 ? (M1 - 1) * 8 : 1;
            f32x2 P[NP];                                                    // P[(s-1)*8 + i] = (phi_s(2i), phi_s(2i+1)), i < NPB/2
            if constexpr (M1 > 1) load_pairs<NP>(P, tab + MOMENT_OFF_PHI, A[0][15]);
            static_for<(HI + 3) / 4 - LO / 4>([&](auto qc) -> bool {
                constexpr int q = LO / 4 + decltype(qc)::value;
                const float4 v = vec(win, q);
                const f32x2 e[2] = {f32x2{v.x, v.y}, f32x2{v.z, v.w}};
                static_for<2>([&](auto hc) -> bool {
                    constexpr int I = 4 * q + 2 * decltype(hc)::value;      // first sample of this aligned pair
                    if constexpr (I >= LO && I < HI) {
                        constexpr int i = (I - LO) / 2;                     // pair i of the block holds t = 2i, 2i+1
                        const f32x2 x = e[decltype(hc)::value];
                        if constexpr (i == 0) M[0] = x; else M[0] += x;
                        static_for<M1 - 1>([&](auto sc) -> bool {
                            constexpr int s = decltype(sc)::value + 1;
                            if constexpr (i == 0) M[s] = pk_mul_pair(P[(s - 1) * 8], x);
                            else if constexpr (i < NPB / 2) pk_fma_pair(M[s], P[(s - 1) * 8 + i], x);
                            else pk_fma_pair_swapped<(s & 1) != 0>(M[s], P[(s - 1) * 8 + NPB - 1 - i], x);    // phi_s(BK-1-t) = (-1)^s phi_s(t)
                            return true;
                        });
                    }
                    return true;
                });
                return true;
            });
#pragma unroll
            for (int s = 0; s < M1; ++s) M[s].x += M[s].y;
        }

        // ---- 5. the block's share of every output, sum_s (c_s(2J), c_s(2J+1)) * mu_s from the highest moment down, then the join ----
        static_for<M1>([&](auto sc) -> bool {
            constexpr int s = M1 - 1 - decltype(sc)::value;
            f32x2 Cs[16];
            load_pairs<16>(Cs, tab + MOMENT_OFF_C + s * 32, M[s]);
#pragma unroll
            for (int J = 0; J < 16; ++J) {
                if constexpr (s == M1 - 1) {
                    A[1][J] = f32x2{0.0f, 0.0f};
                    pk_fma_pair_bcast(A[1][J], Cs[J], M[s]);
                } else pk_fma_pair_bcast(A[1][J], Cs[J], M[s]);
            }
            return true;
        });

#pragma unroll
        for (int j = 0; j < 16; ++j) { const f32x2 a = A[0][j] + A[1][j]; acc[2 * j] = a.x; acc[2 * j + 1] = a.y; }
    }
};

template <int N, int M1>
__global__ __launch_bounds__(256, 4) void sg1d_center_moment_kernel(const Job1D job, const MomentArgs args)
{
    sg1d_tile_body<float, N, MomentConv<N, M1>>(job, args);
}

}  // namespace sg
