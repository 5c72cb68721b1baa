// sg_k1d_momenth.hpp -- fp32 block moments on HALF-lane blocks: half windows 20..32 (MOMENTH_MIN_N) at ~19 packed multiply-adds per output pair at
// n = 32 (round 5; round 2's whole-lane form -- removed in round 6 -- needed 23, the plain sum 33).
//
// Reference loop served: the centre loop of savgol_apply, src/savgolFilter.c:763-766 (convolve_ilp :547-580).
//
// Why.  Round 5's fp64 block-moment kernel (sg_k1d_moment64.hpp) reaches 0.78-0.84 of the HBM roofline where the fp32 headline sits at 0.73-0.76
// -- with FEWER vector instructions per byte (1.13 M against 1.74 M per SIMD and 34 GB): the fp32 kernel still waits on its multiply-adds a third
// of the time (profiles/r05_1d_f32_n32_pmc_summary.json).  What makes the fp64 form cheap is its geometry: 16 outputs share 2n - 14 samples, so
// only 15 taps per output stay direct.  A lane of the fp32 tile owns 32 outputs; here it treats them as TWO groups of 16, each with its own block.
//
// One group: outputs r = 0..15 read X[r + OFF + k].  The samples X[LO .. HI) (LO even >= 15 + OFF, HI even <= OFF + 2n + 1: 48 of 80 at n = 32) lie
// in every output's window; there the taps are a polynomial q_r(t), and  sum_t q_r(t) X[LO+t] = sum_s c_s(r) mu_s  with the block's Legendre moments.
//   * head / tail (samples outside the block, <= 17 taps per output): "x stationary" -- one sample, broadcast to both halves of the instruction, times
//     the SGPR pair (w[k], w[k-1]) feeds the output pair (2j, 2j+1); no straddled input pairs, no moves.  Three round-robin chains as everywhere.
//   * moments: block samples paired front to back, two pairs per instruction: e = front + back, o = front - back, then M1 - 1 multiply-adds
//     (phi_s(BK-1-t) = (-1)^s phi_s(t): even moments from e, odd ones from o);
//   * M1 multiply-adds per output pair for the block's share, added LAST (it is the largest single term of an output that cancels).
// ~300 packed instructions per 16 outputs at n = 32, M1 = 5.  Constants through pinned scalar loads.
#pragma once

#include "sg_k1d.hpp"

namespace sg {

typedef f32x2 __attribute__((address_space(4))) ConstPairH;

// acc += S * (x.HALF, x.HALF): the tap pair as it stands, one sample broadcast
template <int HALF>
__device__ __forceinline__ void pk_fma_xb(f32x2 &acc, const f32x2 s, const f32x2 x)
{
    if constexpr (HALF == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(s), "v"(x));
    else                     asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(s), "v"(x));
}
// (a.x + b.y, a.y + b.x) and (a.x - b.y, a.y - b.x): a front pair with the back pair that mirrors it
__device__ __forceinline__ f32x2 pk_add_swapped(const f32x2 a, const f32x2 b)
{
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_sub_swapped(const f32x2 a, const f32x2 b)
{
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// S * (x.HALF, x.HALF): the first term of a chain (no zero-initialised accumulator)
template <int HALF>
__device__ __forceinline__ f32x2 pk_mul_xb(const f32x2 s, const f32x2 x)
{
    f32x2 p;
    if constexpr (HALF == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "s"(s), "v"(x));
    else                     asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(p) : "s"(s), "v"(x));
    return p;
}
__device__ __forceinline__ f32x2 pk_mul_ss(const f32x2 c, const f32x2 x)
{
    f32x2 p;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "s"(c), "v"(x));
    return p;
}
__device__ __forceinline__ f32x2 pk_mul_cb(const f32x2 c, const f32x2 x)                    // c * x.lo
{
    f32x2 p;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "s"(c), "v"(x));
    return p;
}
__device__ __forceinline__ void pk_fma_ss(f32x2 &acc, const f32x2 c, const f32x2 x)        // acc += c * x, both halves their own
{
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "s"(c), "v"(x));
}
__device__ __forceinline__ void pk_fma_cb(f32x2 &acc, const f32x2 c, const f32x2 x)        // acc += c * x.lo (x.lo broadcast)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(c), "v"(x));
}

template <int N, int M1>
struct MomentHConv {
    typedef K1D<float, N, 8> K;                              // 8 vectors per lane whatever the plain kernel of this half window uses
    typedef MomentArgs Args;
    static constexpr int OFF = K::OFF, LO = momenth_lo(N), HI = momenth_hi(N), BK = HI - LO, STEPS = BK / 4;      // packed steps: two sample pairs each
    static constexpr int WS = 2 * N + 1, CH = 3;
    static_assert(N >= MOMENTH_MIN_N && N <= MOMENT_MAX_N && K::R == 32 && K::VPL == 8, "32 outputs per lane as two groups of 16");
    static_assert(LO % 2 == 0 && HI % 2 == 0 && BK % 4 == 0 && LO >= 15 + OFF && HI <= OFF + 2 * N + 1 && STEPS <= MOMENTH_MAX_STEPS, "block geometry");
    static_assert(M1 >= 1 && M1 <= MOMENT_MAX_TERMS, "1..7 moments");
    // tap pairs (w[k], w[k-1]) the head uses: k = 0 .. LO-1-OFF; the tail: k = HI-OFF-14 .. 2n+1
    static constexpr int HEAD_K1 = LO - 1 - OFF, TAIL_K0 = HI - OFF - 14, TAIL_CNT = 2 * N + 1 - TAIL_K0 + 1;
    static constexpr int LAST = OFF + 2 * N + 15;                     // last sample any of the 16 outputs reads

    template <int COUNT>
    static __device__ __forceinline__ void load_pairs(f32x2 (&dst)[COUNT], const float *p, const f32x2 after)
    {
        asm volatile("" : "+s"(p) : "v"(after));
        const ConstPairH *cp = reinterpret_cast<const ConstPairH *>(reinterpret_cast<uintptr_t>(p));
#pragma unroll
        for (int i = 0; i < COUNT; ++i) dst[i] = cp[i];
    }
    static __device__ __forceinline__ float4 vec(const char *win, int q) { return *reinterpret_cast<const float4 *>(win + slab_vec_off<8>(q)); }

    // one group of 16 outputs whose window starts VB vectors into the lane's window
    template <int VB>
    static __device__ __forceinline__ void group(const char *win, const float *tab, float *out16)
    {
        // chain c of output pair j is opened by the head sample i = OFF + 2j + c (tap pair k = c) with a multiply; where that sample already
        // belongs to the block (the last pair's last chains) the chain starts from zero
        f32x2 A[CH][8];
        static_for<CH * 8>([&](auto ic) -> bool {
            constexpr int c = decltype(ic)::value / 8, j = decltype(ic)::value % 8;
            if constexpr (OFF + 2 * j + c >= LO) A[c][j] = f32x2{0.0f, 0.0f};
            return true;
        });

        // ---- 1. head: samples OFF .. LO-1; sample i is tap k = i - OFF - 2j of output 2j and tap k - 1 of output 2j + 1 ----
        {
            f32x2 W[HEAD_K1 + 1];
            load_pairs<HEAD_K1 + 1>(W, tab + MOMENTH_OFF_W, f32x2{0.0f, 0.0f});
            static_for<(LO + 3) / 4>([&](auto qc) -> bool {
                constexpr int q = decltype(qc)::value;
                const float4 v = vec(win, q + VB);
                const f32x2 x01 = {v.x, v.y}, x23 = {v.z, v.w};
                static_for<4>([&](auto ec) -> bool {
                    constexpr int e = decltype(ec)::value, i = 4 * q + e;
                    if constexpr (i >= OFF && i < LO) {
                        static_for<8>([&](auto jc) -> bool {
                            constexpr int j = decltype(jc)::value, k = i - OFF - 2 * j;
                            if constexpr (k >= 0 && k < CH) A[k][j] = pk_mul_xb<(e & 1)>(W[k], e < 2 ? x01 : x23);
                            else if constexpr (k >= CH && k <= HEAD_K1) pk_fma_xb<(e & 1)>(A[k % CH][j], W[k], e < 2 ? x01 : x23);
                            return true;
                        });
                    }
                    return true;
                });
                return true;
            });
        }
        // ---- 2. tail: samples HI .. LAST ----
        {
            f32x2 W[TAIL_CNT];
            load_pairs<TAIL_CNT>(W, tab + MOMENTH_OFF_W + 2 * TAIL_K0, A[0][0]);
            static_for<LAST / 4 - HI / 4 + 1>([&](auto qc) -> bool {
                constexpr int q = HI / 4 + decltype(qc)::value;
                const float4 v = vec(win, q + VB);
                const f32x2 x01 = {v.x, v.y}, x23 = {v.z, v.w};
                static_for<4>([&](auto ec) -> bool {
                    constexpr int e = decltype(ec)::value, i = 4 * q + e;
                    if constexpr (i >= HI && i <= LAST) {
                        static_for<8>([&](auto jc) -> bool {
                            constexpr int j = decltype(jc)::value, k = i - OFF - 2 * j;
                            if constexpr (k >= TAIL_K0 && k <= 2 * N + 1) pk_fma_xb<(e & 1)>(A[k % CH][j], W[k - TAIL_K0], e < 2 ? x01 : x23);
                            return true;
                        });
                    }
                    return true;
                });
                return true;
            });
        }
        // ---- 3. the taps applied one by one are done: join their chains (the block's share comes last, through a chain of its own) ----
#pragma unroll
        for (int j = 0; j < 8; ++j) A[0][j] = (A[0][j] + A[1][j]) + A[2][j];

        // ---- 4. moments: step u pairs samples (LO + 2u, LO + 2u + 1) with (HI - 1 - 2u, HI - 2 - 2u) ----
        f32x2 M[M1];
        {
            // Two consecutive steps use the two halves of the same front vector and of the same back vector.  Left to itself hipcc narrows every
            // half-used load to 8 bytes and pairs them as ds_read2_b64 -- and 8-byte reads at this slab's 36-dword lane stride meet in banks two ways
            // (lanes l and l + 16): 192 conflict cycles per tile in the counters (31 % of the LDS-active cycles).  So: one 16-byte load per vector,
            // pinned whole by an empty asm, shared by the two steps -- half the LDS reads of this section and none of the conflicts.
            typedef float whole4 __attribute__((ext_vector_type(4)));
            whole4 vf = {0.0f, 0.0f, 0.0f, 0.0f}, vb = {0.0f, 0.0f, 0.0f, 0.0f};
            constexpr int CHUNK = 4;                                         // steps per batch of phi loads (6 pairs per step whatever M1 is)
            static_for<(STEPS + CHUNK - 1) / CHUNK>([&](auto cc) -> bool {
                constexpr int u0 = CHUNK * decltype(cc)::value, nu = STEPS - u0 < CHUNK ? STEPS - u0 : CHUNK;
                f32x2 P[nu * 6];
                if constexpr (M1 > 1) load_pairs<nu * 6>(P, tab + MOMENTH_OFF_PHI + u0 * 12, u0 == 0 ? A[0][7] : M[0]);
                static_for<nu>([&](auto uc) -> bool {
                    constexpr int u = u0 + decltype(uc)::value;
                    constexpr int fi = LO + 2 * u, bi = HI - 2 - 2 * u;        // first sample of the front pair / of the aligned back pair
                    if constexpr (u == 0 || fi % 4 == 0) { vf = *reinterpret_cast<const whole4 *>(win + slab_vec_off<8>(fi / 4 + VB)); asm volatile("" : "+v"(vf)); }
                    if constexpr (u == 0 || bi % 4 == 2) { vb = *reinterpret_cast<const whole4 *>(win + slab_vec_off<8>(bi / 4 + VB)); asm volatile("" : "+v"(vb)); }
                    const f32x2 f = (fi % 4) ? f32x2{vf.z, vf.w} : f32x2{vf.x, vf.y};
                    const f32x2 b = (bi % 4) ? f32x2{vb.z, vb.w} : f32x2{vb.x, vb.y};
                    const f32x2 ev = pk_add_swapped(f, b), od = pk_sub_swapped(f, b);
                    if constexpr (u == 0) M[0] = ev; else M[0] += ev;
                    static_for<M1 - 1>([&](auto sc) -> bool {
                        constexpr int s = decltype(sc)::value + 1;
                        if constexpr (u == 0) M[s] = pk_mul_ss(P[(u - u0) * 6 + (s - 1)], (s & 1) ? od : ev);
                        else pk_fma_ss(M[s], P[(u - u0) * 6 + (s - 1)], (s & 1) ? od : ev);
                        return true;
                    });
                    return true;
                });
                return true;
            });
#pragma unroll
            for (int s = 0; s < M1; ++s) M[s].x += M[s].y;
        }
        // ---- 5. the block's share of every output pair, highest moment first, then the join ----
        static_for<M1>([&](auto sc) -> bool {
            constexpr int s = M1 - 1 - decltype(sc)::value;
            f32x2 Cs[8];
            load_pairs<8>(Cs, tab + MOMENTH_OFF_C + s * 16, M[s]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (s == M1 - 1) A[1][j] = pk_mul_cb(Cs[j], M[s]);
                else pk_fma_cb(A[1][j], Cs[j], M[s]);
            }
            return true;
        });
#pragma unroll
        for (int j = 0; j < 8; ++j) { const f32x2 a = A[0][j] + A[1][j]; out16[2 * j] = a.x; out16[2 * j + 1] = a.y; }
    }

    static __device__ __forceinline__ void run(const char *win, const MomentArgs &args, float (&acc)[32], unsigned)
    {
        group<0>(win, args.table, &acc[0]);
        group<4>(win, args.table, &acc[16]);          // outputs 16..31: the same window 16 samples = 4 vectors further
    }
};

template <int N, int M1>
__global__ __launch_bounds__(64 * SG_K1D_WAVES, 4) void sg1d_center_momenth_kernel(const Job1D job, const MomentArgs args)
{
    sg1d_tile_body<float, N, MomentHConv<N, M1>>(job, args);
}

}  // namespace sg
