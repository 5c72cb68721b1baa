// sg_k1d_moment64.hpp -- the fp64 inner product for half windows 24..32 with block moments: ~36 multiply-adds per output instead of
// 2n + 1 = 65 (round 5; OPT-IN: SAVGOL_BATCH_MOMENT_F64).
//
// Reference loop served: the centre loop of savgol_apply, src/savgolFilter.c:763-766 (convolve_ilp :547-580), on fp64 data -- which
// the reference does not have; the contract is SURVEY 8c's fp64 oracle (the fp32 tables promoted, double accumulation).
//
// Why.  sg1d_center_kernel<double, 32> issues 1040 v_fma_f64 per tile and lane and is issue-bound (4.4 cycles per instruction, 47 % of
// the wave cycles stalled on issue: profiles/r04_1d_f64_n32_pmc_summary.json) at 0.68 of the HBM roofline.  Only fewer instructions help.
//
// How (round 2's block-moment construction, re-derived for 16 outputs per lane and scalar doubles).  A lane owns outputs r = 0..15; its
// window is X[0 .. 16 + 2n + OFF) and output r reads X[r + OFF + k] with tap k.  The samples X[LO .. HI), LO = 15 + OFF, HI = OFF + 2n + 1
// (50 of the 80 at n = 32) lie inside EVERY output's window, and there the taps are a polynomial q_r(t) of degree < M1 in t (the centre
// taps of a Savitzky-Golay filter are samples of a polynomial of degree <= poly_order, reference compute_weight :336-356).  In the
// Legendre basis phi_s of the block:  sum_t q_r(t) X[LO+t] = sum_s c_s(r) mu_s,  mu_s = sum_t phi_s(t) X[LO+t].
//   * head (samples below LO) and tail (from HI on): 15 taps per output, applied one by one -- taps 0..14 only, the tail's through the
//     (anti)symmetry tap[2n-k] = +-tap[k] that the plain fp64 kernel uses as well (sign of the input flipped for odd derivatives);
//   * moments: the block's samples are paired front to back (phi_s(BK-1-t) = (-1)^s phi_s(t)): per pair one sum, one difference and
//     M1 - 1 multiply-adds;
//   * M1 multiply-adds per output for the block's share.
// 240 + 25 (M1 + 2) + 16 M1 = 495 instructions at n = 32, M1 = 5, instead of 1040.  All constants come through scalar loads.
//
// Accuracy.  The block's share uses the POLYNOMIAL fitted to the fp32 table in double (sg1d_moment64_prepare), not the table's own
// fp32-rounded entries: the result differs from the promoted-table oracle by the fit's residual (<= 3e-7 of the largest tap, the rounding
// already in the reference's table; measured ~1e-7 normwise) -- inside north_star's 1e-6, outside the default path's 1e-12.  Hence opt-in.
#pragma once

#include "sg_k1d.hpp"

namespace sg {

typedef double __attribute__((address_space(4))) ConstD64;

template <int N, int M1>
struct Moment64Conv {
    typedef K1D<double, N, 8> K;
    typedef Moment64Args Args;
    static constexpr int OFF = K::OFF, LO = moment64_lo(N), HI = moment64_hi(N), BK = HI - LO, NPAIR = BK / 2;
    static constexpr int S = LO + HI - 1;                      // a block sample i pairs with sample S - i
    static_assert(N >= MOMENT_MIN_N && N <= MOMENT_MAX_N && K::R == 16 && K::VPL == 8 && K::E == 2, "16 outputs per lane, 8 vectors per lane");
    static_assert(OFF == moment64_off(N) && BK % 2 == 0 && BK >= 2 && NPAIR <= MOMENT64_MAX_PAIRS && (S & 1) == 1, "block geometry");
    static_assert(M1 >= 1 && M1 <= MOMENT_MAX_TERMS, "1..7 moments");

    // COUNT doubles from the table through scalar loads, pinned behind `after` (hoisted to the top of the kernel they would all be live)
    template <int COUNT>
    static __device__ __forceinline__ void load_d(double (&dst)[COUNT], const double *p, const double after)
    {
        asm volatile("" : "+s"(p) : "v"(after));
        const ConstD64 *cp = reinterpret_cast<const ConstD64 *>(reinterpret_cast<uintptr_t>(p));
#pragma unroll
        for (int i = 0; i < COUNT; ++i) dst[i] = cp[i];
    }
    static __device__ __forceinline__ double2 vec(const char *win, int q) { return *reinterpret_cast<const double2 *>(win + slab_vec_off<8>(q)); }
    static __device__ __forceinline__ double flipped(double x, unsigned flip) { return __hiloint2double(__double2hiint(x) ^ (int)flip, __double2loint(x)); }

    static __device__ __forceinline__ void run(const char *win, const Moment64Args &args, double (&acc)[16], unsigned flags)
    {
        const double *tab = args.table;
        const unsigned flip = (flags & JOB_ODD_TAPS) ? 0x80000000u : 0u;
        double A[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) A[r] = 0.0;

        // ---- 1. head and tail: the 15 taps per output that fall outside the block, taps 0..14 in SGPR pairs ----
        {
            double W[15];
            load_d<15>(W, tab + MOMENT64_OFF_W, 0.0);
            // head: sample i < LO is tap k = i - OFF - r of output r, k = 0..14
            static_for<(LO + 1) / 2>([&](auto qc) -> bool {
                constexpr int q = decltype(qc)::value;
                const double2 v = vec(win, q);
                static_for<2>([&](auto ec) -> bool {
                    constexpr int i = 2 * q + decltype(ec)::value;
                    if constexpr (i >= OFF && i < LO) {
                        const double x = decltype(ec)::value ? v.y : v.x;
                        static_for<16>([&](auto rc) -> bool {
                            constexpr int r = decltype(rc)::value, k = i - OFF - r;
                            if constexpr (k >= 0 && k <= 14) A[r] = __builtin_fma(W[k], x, A[r]);
                            return true;
                        });
                    }
                    return true;
                });
                return true;
            });
            // tail: sample i = HI - 1 + j (j = 1..15) is tap 2n + j - r of output r >= j, i.e. +-tap r - j
            static_for<(HI + 15) / 2 - HI / 2 + 1>([&](auto qc) -> bool {
                constexpr int q = HI / 2 + decltype(qc)::value;
                if constexpr (q < K::WQ) {
                    const double2 v = vec(win, q);
                    static_for<2>([&](auto ec) -> bool {
                        constexpr int i = 2 * q + decltype(ec)::value, j = i - (HI - 1);
                        if constexpr (j >= 1 && j <= 15) {
                            const double xs = flipped(decltype(ec)::value ? v.y : v.x, flip);
                            static_for<16>([&](auto rc) -> bool {
                                constexpr int r = decltype(rc)::value, km = r - j;
                                if constexpr (km >= 0 && km <= 14) A[r] = __builtin_fma(W[km], xs, A[r]);
                                return true;
                            });
                        }
                        return true;
                    });
                }
                return true;
            });
        }

        // ---- 2. moments of the block, samples paired front to back: pair t = (X[LO + t], X[HI - 1 - t]), t < BK / 2 ----
        double M[M1];
#pragma unroll
        for (int s = 0; s < M1; ++s) M[s] = 0.0;
        {
            // a front vector q holds samples 2q, 2q + 1; their partners S - 2q, S - 2q - 1 are the .y and .x of vector (S - 1) / 2 - q
            constexpr int Q0 = LO / 2, Q1 = (LO + NPAIR - 1) / 2;                 // front vectors Q0 .. Q1
            constexpr int CHUNK = 2;                                               // front vectors per batch of phi loads (4 pairs x 6 doubles)
            static_for<(Q1 - Q0 + CHUNK) / CHUNK>([&](auto cc) -> bool {
                constexpr int qa = Q0 + CHUNK * decltype(cc)::value;
                // phi of pairs t0 .. t0 + 2 CHUNK - 1 (the table holds 6 doubles per pair whatever M1 is), t0 = 2 qa - LO (may be -1: clamp)
                constexpr int t0 = 2 * qa - LO < 0 ? 0 : 2 * qa - LO;
                constexpr int nt = (2 * (qa + CHUNK) - LO > NPAIR ? NPAIR : 2 * (qa + CHUNK) - LO) - t0;
                double P[(nt > 0 ? nt : 1) * 6];
                if constexpr (M1 > 1 && nt > 0) load_d<nt * 6>(P, tab + MOMENT64_OFF_PHI + t0 * 6, M[0]);
                static_for<CHUNK>([&](auto uc) -> bool {
                    constexpr int q = qa + decltype(uc)::value;
                    if constexpr (q <= Q1) {
                        const double2 vf = vec(win, q), vb = vec(win, (S - 1) / 2 - q);
                        static_for<2>([&](auto ec) -> bool {
                            constexpr int i = 2 * q + decltype(ec)::value, t = i - LO;
                            if constexpr (t >= 0 && t < NPAIR) {
                                const double xf = decltype(ec)::value ? vf.y : vf.x;       // X[i]
                                const double xb = decltype(ec)::value ? vb.x : vb.y;       // X[S - i]
                                const double e = xf + xb, o = xf - xb;
                                M[0] += e;
                                static_for<M1 - 1>([&](auto sc) -> bool {
                                    constexpr int s = decltype(sc)::value + 1;
                                    M[s] = __builtin_fma(P[(t - t0) * 6 + (s - 1)], (s & 1) ? o : e, M[s]);
                                    return true;
                                });
                            }
                            return true;
                        });
                    }
                    return true;
                });
                return true;
            });
        }

        // ---- 3. the block's share of every output: sum_s c_s(r) mu_s, highest moment first ----
        static_for<M1>([&](auto sc) -> bool {
            constexpr int s = M1 - 1 - decltype(sc)::value;
            double Cs[16];
            load_d<16>(Cs, tab + MOMENT64_OFF_C + s * 16, M[s]);
#pragma unroll
            for (int r = 0; r < 16; ++r) A[r] = __builtin_fma(Cs[r], M[s], A[r]);
            return true;
        });
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = A[r];
    }
};

template <int N, int M1>
__global__ __launch_bounds__(256, 3) void sg1d_center_moment64_kernel(const Job1D job, const Moment64Args args)
{
    sg1d_tile_body<double, N, Moment64Conv<N, M1>>(job, args);
}

}  // namespace sg
