// sg_k1d_moment.hpp -- the half_window = 32 fp32 inner product with 43 multiply-adds per output instead of 65.
//
// Why.  At n = 32 sg1d_center_kernel is not limited by HBM but by the energy of its 65 FMAs per sample: the vector unit is
// issue-saturated (4.25 cycles per VALU instruction per SIMD, PMC) at the 1.5 GHz the chip holds under this load
// (profiles/r02_*).  Removing index arithmetic did not help (-10 % instructions, -0.6 % time): only fewer
// FMAs do.
//
// How.  Savitzky-Golay centre taps are samples of a polynomial of degree <= poly_order in the tap index k
// (reference compute_weight, src/savgolFilter.c:336-356: a sum of Gram polynomials F_j(k)).  A lane owns 32 consecutive
// outputs r = 0..31; its window is samples X[0..95] and output r reads X[r .. r+64] with tap k = i - r on sample i.  The
// lane's OWN block X[32..63] lies inside the window of every one of its outputs, and on it the taps w[t - r + 32]
// (t = i - 32) are a polynomial q_r(t) of degree < M1.  Written in a basis phi_s(t) (Legendre polynomials on the block),
//     sum_t q_r(t) X[32+t]  =  sum_{s<M1} c_s(r) * mu_s,      mu_s = sum_t phi_s(t) X[32+t]     (block moments, once per lane)
// so the 32 taps that fall on the own block cost M1 multiply-adds per output plus M1 per sample for the moments, and only
// the 33 taps outside it (X[r..31] and X[64..64+r]) are applied one by one: 33 + 2*M1 = 43 at poly_order 4 instead of 65.
// The moments are LOCAL (one 32-sample block, no recurrence, no running sums), which is why fp32 holds: measured
// normwise error vs the fp64 oracle 3.0e-7 ... 4.5e-7 for smoothing filters, the same as the plain sum (tests/test_gpu_1d.py).
//
// The host (sg1d_moment_prepare, sg_api_1d.cpp) fits the polynomial to the filter's fp32 table in double, refuses tables
// that are not a polynomial to 3e-7 of max|w| (hand-edited tables run the plain kernel), and uploads
//     w[66] | phi[s-1][t], s = 1..6, t = 0..15 | c[s][J] = (c_s(2J), c_s(2J+1)), s = 0..6, J = 0..15
// (MomentTable in sg_k1d_host.hpp).  Everything is read through scalar loads and lives in SGPRs.
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

template <int M1>
struct MomentConv {
    typedef K1D<float, 32> K;
    typedef MomentArgs Args;
    static_assert(K::R == 32 && K::OFF == 0 && K::NA == 32 && K::VPL == 8, "written for 32 outputs per lane, halo 32");
    static_assert(M1 >= 1 && M1 <= MOMENT_MAX_TERMS, "1..7 moments");

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

    // ---- taps outside the own block, input stationary: the pair (X[I], X[I+1]) feeds every accumulator pair J it touches ----
    template <int I, int J = 0>
    static __device__ __forceinline__ void feed_head(f32x2 (&A)[16], const f32x2 (&W)[16], const f32x2 x)      // I + 1 <= 31, taps 0..30
    {
        if constexpr (J < 16) {
            constexpr int k = I - 2 * J;
            if constexpr (k == 0) A[J] = pk_mul_sgpr<0>(W[0], x);
            else if constexpr (k > 0) pk_fma_sgpr<(k & 1)>(A[J], W[k >> 1], x);
            feed_head<I, J + 1>(A, W, x);
        }
    }
    template <int I, int J = 0>
    static __device__ __forceinline__ void feed_tail(f32x2 (&A)[16], const f32x2 (&W)[17], const f32x2 x)      // I >= 64, taps 34..64 (W[p] = taps 32+2p, 33+2p)
    {
        if constexpr (J < 16) {
            constexpr int k = I - 2 * J;
            if constexpr (k <= 64) pk_fma_sgpr<(k & 1)>(A[J], W[(k - 32) >> 1], x);
            feed_tail<I, J + 1>(A, W, x);
        }
    }

    static __device__ __forceinline__ float4 vec(const char *win, int q) { return *reinterpret_cast<const float4 *>(win + slab_vec_off<8>(q)); }

    static __device__ __forceinline__ void run(const char *win, const MomentArgs &args, float (&acc)[32], unsigned)
    {
        const float *tab = args.table;
        f32x2 A[16];

        // ---- 1. head: X[0..31], taps k = i - r for i < 32 ----
        {
            f32x2 W[16];
            load_pairs<16>(W, tab + MOMENT_OFF_W, f32x2{0.0f, 0.0f});
            f32x2 prev = f32x2{0.0f, 0.0f};
            static_for<8>([&](auto qc) -> bool {
                constexpr int q = decltype(qc)::value;
                const float4 v = vec(win, q);
                const f32x2 e0 = {v.x, v.y}, e1 = {v.z, v.w};
                if constexpr (q > 0) feed_head<4 * q - 1>(A, W, pk_straddle(prev, e0));
                feed_head<4 * q>(A, W, e0);
                feed_head<4 * q + 1>(A, W, pk_straddle(e0, e1));
                feed_head<4 * q + 2>(A, W, e1);
                prev = e1;
                return true;
            });
            // X[31] only reaches the even output of each pair (tap 31 - 2J); its partner X[32] is in the own block
            const float x31 = prev.y;
#pragma unroll
            for (int J = 0; J < 16; ++J) A[J].x = __builtin_fmaf((31 - 2 * J) & 1 ? W[(31 - 2 * J) >> 1].y : W[(31 - 2 * J) >> 1].x, x31, A[J].x);
        }

        // ---- 2. moments of the own block X[32..63]: M[s] = (sum over even t, sum over odd t) of phi_s(t) X[32+t] ----
        f32x2 M[M1];
        {
            constexpr int NP = M1 > 1 ? (M1 - 1) * 8 : 1;
            f32x2 P[NP];                                                    // P[(s-1)*8 + i] = (phi_s(2i), phi_s(2i+1)), i < 8
            if constexpr (M1 > 1) load_pairs<NP>(P, tab + MOMENT_OFF_PHI, A[15]);
            static_for<8>([&](auto qc) -> bool {
                constexpr int q = decltype(qc)::value;
                const float4 v = vec(win, 8 + q);
                const f32x2 e[2] = {f32x2{v.x, v.y}, f32x2{v.z, v.w}};
                static_for<2>([&](auto hc) -> bool {
                    constexpr int i = 2 * q + decltype(hc)::value;          // pair i holds t = 2i, 2i+1
                    const f32x2 x = e[decltype(hc)::value];
                    if constexpr (i == 0) M[0] = x; else M[0] += x;
                    static_for<M1 - 1>([&](auto sc) -> bool {
                        constexpr int s = decltype(sc)::value + 1;
                        if constexpr (i == 0) M[s] = pk_mul_pair(P[(s - 1) * 8], x);
                        else if constexpr (i < 8) pk_fma_pair(M[s], P[(s - 1) * 8 + i], x);
                        else pk_fma_pair_swapped<(s & 1) != 0>(M[s], P[(s - 1) * 8 + 15 - i], x);    // phi_s(31 - t) = (-1)^s phi_s(t)
                        return true;
                    });
                    return true;
                });
                return true;
            });
#pragma unroll
            for (int s = 0; s < M1; ++s) M[s].x += M[s].y;
        }

        // ---- 3. the own block's share of every output: A[J] += (c_s(2J), c_s(2J+1)) * mu_s ----
        static_for<M1>([&](auto sc) -> bool {
            constexpr int s = decltype(sc)::value;
            f32x2 Cs[16];
            load_pairs<16>(Cs, tab + MOMENT_OFF_C + s * 32, M[s]);
#pragma unroll
            for (int J = 0; J < 16; ++J) pk_fma_pair_bcast(A[J], Cs[J], M[s]);
            return true;
        });

        // ---- 4. tail: X[64..95], taps k = i - r for i >= 64 ----
        {
            f32x2 W[17];
            load_pairs<17>(W, tab + MOMENT_OFF_W + 32, A[0]);
            f32x2 prev = f32x2{0.0f, 0.0f};
            static_for<8>([&](auto qc) -> bool {
                constexpr int q = decltype(qc)::value;
                const float4 v = vec(win, 16 + q);
                const f32x2 e0 = {v.x, v.y}, e1 = {v.z, v.w};
                if constexpr (q == 0) {
                    // X[64] also reaches the odd output of each pair on its own (tap 63 - 2J); its partner X[63] is in the own block
#pragma unroll
                    for (int J = 0; J < 16; ++J) A[J].y = __builtin_fmaf((63 - 2 * J) & 1 ? W[(31 - 2 * J) >> 1].y : W[(31 - 2 * J) >> 1].x, v.x, A[J].y);
                } else {
                    feed_tail<64 + 4 * q - 1>(A, W, pk_straddle(prev, e0));
                }
                feed_tail<64 + 4 * q>(A, W, e0);
                feed_tail<64 + 4 * q + 1>(A, W, pk_straddle(e0, e1));
                feed_tail<64 + 4 * q + 2>(A, W, e1);
                prev = e1;
                return true;
            });
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc[2 * j] = A[j].x; acc[2 * j + 1] = A[j].y; }
    }
};

template <int M1>
__global__ __launch_bounds__(256, 4) void sg1d_center_moment_kernel(const Job1D job, const MomentArgs args)
{
    sg1d_tile_body<float, 32, MomentConv<M1>>(job, args);
}

}  // namespace sg
