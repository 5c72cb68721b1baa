// sg_k1d.hpp -- the 1-D sliding-window convolution kernel for gfx950 (CDNA4), templated on the
// sample type and on the half window N (compile-time so that every tap index is a literal).
//
// Reference loop replaced: savgol_apply centre loop, src/savgolFilter.c:763-766 (+ the index
// remaps of get_padded_sample :442-482 for the non-polynomial boundary modes, + savgol_apply_valid
// :843-847 through the store window).  The POLYNOMIAL edge rows are a separate tiny kernel
// (sg1d_edge_item below, extra items of the same launch; reference :773-784).
//
// Mapping (one 64-lane wave = one tile, waves never talk to each other, no s_barrier):
//   * a tile is 64*R consecutive outputs of one channel, R = 128 B / sizeof(T) (32 fp32, 16 fp64);
//   * the wave copies tile + halo from HBM into its private LDS slab with coalesced 16-B loads
//     (a wave handles one tile; the other waves of the SIMD cover its load latency);
//   * lane l then owns outputs [l*R, l*R+R): it walks its R+2N inputs once with ds_read_b128 and
//     feeds each input into every accumulator it touches ("input stationary"): R accumulators in
//     VGPRs, the 2N+1 taps in SGPRs (they arrive as a by-value kernel argument), one v_fmac per
//     tap per output, no shuffles, no LDS traffic inside the inner product;
//   * results go back through the slab so that the global stores are coalesced 16-B rows.
// The slab is padded by 16 B every 128 B, which makes the lane stride 144 B = 36 banks: the per-lane b128 window reads and the row-wise
// b128 staging writes are bank-conflict free (MI355X_MICROARCH.md, LDS table).  The RESULTS take another layout on their way out (round 5):
// read back row by row through the padded layout, every ds_read_b128 lane group met two pads and lost a cycle -- exactly 32 conflict cycles
// per tile, 10.8 % of the LDS-active cycles (profiles/r04_1d_f32_n32_pmc_summary.json) -- so they are written unpadded with an XOR swizzle
// of the vector index (result_vec_off), conflict free for the per-lane writes and for the row-wise reads alike.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sg_k1d_host.hpp"
#include "sg_pk.hpp"

namespace sg {

#ifdef SG_STAMPS   // diagnostic build only (tools/stamp_1d.hip): phase timestamps of block 0 / wave 0
__device__ unsigned long long *g_stamps;
__device__ __forceinline__ void stamp(bool on, int it, int slot)
{
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (on && it < 64 && (threadIdx.x & 63) == 0) g_stamps[it * 8 + slot] = t;
}
#define SG_STAMP(slot) stamp(stamp_on, stamp_it, slot)
#else
#define SG_STAMP(slot) do {} while (0)
#endif

template <typename T> struct V16;
template <> struct V16<float>  { typedef float4  type; static constexpr int E = 4; };
template <> struct V16<double> { typedef double2 type; static constexpr int E = 2; };

__device__ __forceinline__ float  vget(const float4 &v, int e)  { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
__device__ __forceinline__ double vget(const double2 &v, int e) { return e == 0 ? v.x : v.y; }
__device__ __forceinline__ void   vset(float4 &v, int e, float x)  { if (e == 0) v.x = x; else if (e == 1) v.y = x; else if (e == 2) v.z = x; else v.w = x; }
__device__ __forceinline__ void   vset(double2 &v, int e, double x) { if (e == 0) v.x = x; else v.y = x; }

__device__ __forceinline__ float  fma_t(float a, float b, float c)    { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// streamed-once data: nontemporal loads/stores (SG_NT=0 falls back to the default cache policy)
#ifndef SG_NT
#define SG_NT 1
#endif
template <typename VT> __device__ __forceinline__ VT ld_stream(const VT *p)
{
    static_assert(sizeof(VT) == 16, "16-byte vectors only");
#if SG_NT
    const u32x4 raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return __builtin_bit_cast(VT, raw);
#else
    return *p;
#endif
}
template <typename VT> __device__ __forceinline__ void st_stream(VT *p, const VT &v)
{
    static_assert(sizeof(VT) == 16, "16-byte vectors only");
#if SG_NT
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4 *>(p));
#else
    *p = v;
#endif
}

// index remap of the padded boundary modes (reference get_padded_sample, savgolFilter.c:452-476)
__device__ __forceinline__ int remap_index(int i, int L, int mode, bool &zero)
{
    zero = false;
    if (mode == SAVGOL_BOUNDARY_REFLECT) {
        if (i < 0) { i = -i - 1; if (i >= L) i = L - 1; }
        else       { i = 2 * L - i - 1; if (i < 0) i = 0; }
    } else if (mode == SAVGOL_BOUNDARY_PERIODIC) {
        i = ((i % L) + L) % L;
    } else if (mode == SAVGOL_BOUNDARY_CONSTANT) {
        i = (i < 0) ? 0 : L - 1;
    } else {
        zero = true; i = 0;
    }
    return i;
}

// half windows from which the fp32 inner products keep two partial sums per output (below: one chain of <= 2N+1 <= 7 terms)
#ifndef SG_CHAIN_SPLIT_MIN_N
#define SG_CHAIN_SPLIT_MIN_N 4
#endif

template <typename T, int N, int V = vectors_per_lane(sizeof(T), N)>
struct K1D {
    typedef typename V16<T>::type VT;
    static constexpr int E    = V16<T>::E;
    static constexpr int VPL  = V;                           // 16-B vectors of output per lane (sg_k1d_host.hpp: narrow / wide tiles)
    static constexpr int R    = VPL * E;                     // outputs per lane
    static constexpr int TV   = 64 * VPL;                    // vectors of output per tile
    static constexpr int TW   = 64 * R;                      // outputs per tile
    static constexpr int NA   = (N + E - 1) / E * E;         // halo rounded to whole vectors
    static constexpr int OFF  = NA - N;
    static constexpr int HV   = NA / E;                      // halo vectors per side
    static constexpr int SV   = TV + 2 * HV;                 // vectors in a slab
    static constexpr int SL   = SV * E;                      // elements in a slab
    static constexpr int WQ   = (NA + R + N + E - 1) / E;    // vectors a lane reads
    static constexpr int SLAB = 16 * (SV + (SV + VPL - 1) / VPL);   // bytes: one pad vector after every VPL
#ifndef SG_K1D_WAVES
#define SG_K1D_WAVES 4
#endif
    static constexpr int WAVES = SG_K1D_WAVES;               // waves per block, each with its own slab (A/B builds of one kernel family override; the host counts blocks of 4 tiles)
    // waves per SIMD the register allocation must allow (the LDS slabs allow as many blocks per CU)
    // (fp64 fits 128 VGPRs since its taps moved to SGPRs, but A/B'd in one process the 168-VGPR schedule is 2.5 % faster)
    // (the 12 / 16 KiB tiles hold 3 / 2 blocks per CU in LDS: asking for more only caps the registers for nothing)
    static constexpr int MIN_WAVES = VPL >= 16 ? 2 : VPL >= 12 ? 3 : sizeof(T) == 8 ? 3 : (VPL <= 4 ? 7 : VPL <= 6 ? 5 : 4);
    static_assert(2 * HV <= 64, "halo must fit one extra vector per lane");
    static_assert(WQ <= SV - VPL * 63, "lane 63's window must stay inside the slab");
    static_assert(VPL == 4 || VPL == 6 || VPL == 8 || VPL == 12 || VPL == 16, "lane stride (VPL+1)*16 B must be conflict free for ds_read_b128");
};

// byte offset of slab vector v: one 16-B pad after every VPL vectors, so a lane's VPL vectors are contiguous and
// the lane stride is (VPL+1)*16 B = 20 / 28 / 36 banks -- conflict free for the ds_read_b128 lane groups
template <int VPL> __device__ __forceinline__ constexpr int slab_vec_off(int v) { return 16 * (v + v / VPL); }

// byte offset of RESULT vector v (lane v / VPL's s-th vector, s = v % VPL) for 8 vectors per lane: no pads, the low three bits of the vector
// index XORed with the owner lane's low three bits.  Per-lane writes (ds_write_b128: groups of 8 consecutive lanes, 32 banks) land on
// 4 * (s ^ lane % 8): eight different quads of banks.  Row-wise reads (ds_read_b128: the lane groups {0-3, 12-15, 20-27}, ..., 64 banks) of
// vector p = lane + 64 s' land on 16 different quads as well (worked out in profiles/EXPERIMENTS.md R5.3).
__device__ __forceinline__ constexpr int result_vec_off8(int v) { return 16 * ((v & ~7) | ((v & 7) ^ ((v >> 3) & 7))); }


// ---------------------------------------------------------------------------------------------
// The inner product, "input stationary": walk the lane's window once, feed every input into all
// the accumulators it touches.  acc[r] = sum_k w[k] * x[r + k + OFF], k ascending, one FMA chain.
// ---------------------------------------------------------------------------------------------
template <typename T, int N, int V>
struct Conv {                      // generic form: one v_fma per tap per output
    typedef K1D<T, N, V> K;
    typedef typename K::VT VT;
    static __device__ __forceinline__ void run(const char *win, const Taps &taps, T (&acc)[K::R])
    {
        constexpr int E = K::E, R = K::R, OFF = K::OFF;
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = T(0);
#pragma unroll
        for (int q = 0; q < K::WQ; ++q) {
            const VT v = *reinterpret_cast<const VT *>(win + slab_vec_off<K::VPL>(q));
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const T x = vget(v, e);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int k = q * E + e - r - OFF;             // literal after unrolling
                    if (k >= 0 && k <= 2 * N) acc[r] = fma_t((T)taps.w[k], x, acc[r]);
                }
            }
        }
    }
};

// fp32: plain v_fma_f32 runs at half the packed rate on CDNA4 (a wave64 VALU op takes 4 cycles either
// way), so the taps loop is written with v_pk_fma_f32: accumulators live as pairs (acc[2j], acc[2j+1]),
// inputs as pairs (x[i], x[i+1]), and one instruction does  pair_j += w[k] * (x[i], x[i+1])  with
// i = 2j + k + OFF.  The tap is broadcast out of an aligned SGPR pair with op_sel, so the 65 taps still
// occupy 66 SGPRs; pairs that start at an odd i are assembled with one v_pk_mov_b32 each and then
// shared by up to 16 accumulator pairs (sg_pk.hpp has the instructions).
template <int N, int V>
struct Conv<float, N, V> {
    typedef K1D<float, N, V> K;
    // Round 4 (VERDICT r03 missing #2): THREE partial sums per output, tap k in chain k mod 3, joined once at the end.
    // The reference sums in four round-robin chains of (2N+1)/4 products, each product and each add rounded (src/savgolFilter.c:547-580);
    // ONE chain of 2N+1 fused multiply-adds was up to 3.1 x the reference's own error on derivative filters
    // (profiles/r03_fp32_accuracy_sweep.txt).  What matters is not the chain length alone: two chains of contiguous halves are no
    // better than one (each half of an antisymmetric window sums to something large that the join cancels), while chains INTERLEAVED
    // over the window each look like the whole filter at a coarser step.  Emulated in fp32 on the test's signals, worst case over
    // poly_order <= 6, derivative <= 2 as a multiple of the reference's own error (profiles/r04_fp32_chains.txt): one chain 1.5-3.6,
    // two halves 1.6-4.0, even/odd 0.8-1.8, three round robin 0.7-1.1, four round robin (the reference's split, fused) 0.8-1.0.
    // Cost of three: R more accumulator registers and R packed adds per lane (+3 % instructions at n = 32, time within noise).
    static constexpr int CH = N >= SG_CHAIN_SPLIT_MIN_N ? 3 : 1;
    // all accumulator pairs fed by the input pair that starts at window index I
    template <int I, int J = 0>
    static __device__ __forceinline__ void feed(f32x2 (&A)[CH][K::R / 2], const f32x2 (&W)[33], const f32x2 x)
    {
        if constexpr (J < K::R / 2) {
            constexpr int k = I - 2 * J - K::OFF;
            if constexpr (k >= 0 && k < CH) A[k % CH][J] = pk_mul_sgpr<(k & 1)>(W[k >> 1], x);      // first term of a chain: no zero-initialised accumulator
            else if constexpr (k >= CH && k <= 2 * N) pk_fma_sgpr<(k & 1)>(A[k % CH][J], W[k >> 1], x);
            feed<I, J + 1>(A, W, x);
        }
    }
    template <int Q>
    static __device__ __forceinline__ void quads(const char *win, f32x2 (&A)[CH][K::R / 2], const f32x2 (&W)[33], f32x2 prev)
    {
        if constexpr (Q < K::WQ) {
            const float4 v = *reinterpret_cast<const float4 *>(win + slab_vec_off<K::VPL>(Q));
            const f32x2 e0 = {v.x, v.y}, e1 = {v.z, v.w};
            // pairs that start at an odd index are assembled from the registers already loaded (one v_pk_mov_b32 each);
            // reading them from LDS instead (ds_read2_b32) was measured 4 % slower
            if constexpr (Q > 0) feed<4 * Q - 1>(A, W, pk_straddle(prev, e0));
            feed<4 * Q>(A, W, e0);
            feed<4 * Q + 1>(A, W, pk_straddle(e0, e1));
            feed<4 * Q + 2>(A, W, e1);
            quads<Q + 1>(win, A, W, e1);
        }
    }
    static __device__ __forceinline__ void run(const char *win, const Taps &taps, float (&acc)[K::R])
    {
        f32x2 W[33];
#pragma unroll
        for (int p = 0; p < 33; ++p) W[p] = f32x2{taps.w[2 * p], taps.w[2 * p + 1]};
        f32x2 A[CH][K::R / 2];
        quads<0>(win, A, W, f32x2{0.0f, 0.0f});
#pragma unroll
        for (int j = 0; j < K::R / 2; ++j) {
            f32x2 a = A[0][j];
            if constexpr (CH == 3) a = (a + A[1][j]) + A[2][j];
            acc[2 * j] = a.x; acc[2 * j + 1] = a.y;
        }
    }
};

// fp64: 65 doubles do not fit the scalar register file, but the centre taps savgol_create builds are (anti)symmetric
// bit for bit (tap[2N-k] = +-tap[k]: only Gram terms of the derivative's parity are non-zero at t = 0), so the host
// passes taps 0..N as doubles (33 SGPR pairs at N = 32, exact promotions of the fp32 table) and the kernel applies the
// mirrored half through the sign of the input.  No conversion and no tap in a VGPR inside the loop.
template <int N, int V>
struct Conv<double, N, V> {
    typedef K1D<double, N, V> K;
    // taps 0..N as doubles in SGPR pairs (33 x 2 SGPRs at N = 32); tap 2N-k = +-tap k is applied by flipping the
    // sign of the input instead (xs = +-x, one v_xor_b32 per input), so v_fmac_f64 takes the tap straight from SGPRs
    template <int I, int Rr = 0>
    static __device__ __forceinline__ void feed(double (&acc)[K::R], const Taps &taps, const double x, const double xs)
    {
        if constexpr (Rr < K::R) {
            constexpr int k = I - Rr - K::OFF;
            if constexpr (k >= 0 && k <= N) acc[Rr] = __builtin_fma(taps.wd[k], x, acc[Rr]);
#ifdef SG_F64_HALF_ADDS    // timing experiment only (wrong results): the instruction mix of the symmetric-tap fold, 32 adds + 33 multiply-adds
            else if constexpr (k > N && k <= 2 * N) acc[Rr] = acc[Rr] + xs;
#else
            else if constexpr (k > N && k <= 2 * N) acc[Rr] = __builtin_fma(taps.wd[2 * N - k], xs, acc[Rr]);
#endif
            feed<I, Rr + 1>(acc, taps, x, xs);
        }
    }
    template <int Q>
    static __device__ __forceinline__ void vecs(const char *win, double (&acc)[K::R], const Taps &taps, const unsigned flip)
    {
        if constexpr (Q < K::WQ) {
            const double2 v = *reinterpret_cast<const double2 *>(win + slab_vec_off<K::VPL>(Q));
            const double sx = __hiloint2double(__double2hiint(v.x) ^ (int)flip, __double2loint(v.x));
            const double sy = __hiloint2double(__double2hiint(v.y) ^ (int)flip, __double2loint(v.y));
            feed<2 * Q>(acc, taps, v.x, sx);
            feed<2 * Q + 1>(acc, taps, v.y, sy);
            vecs<Q + 1>(win, acc, taps, flip);
        }
    }
    static __device__ __forceinline__ void run(const char *win, const Taps &taps, double (&acc)[K::R], unsigned flip)
    {
#pragma unroll
        for (int r = 0; r < K::R; ++r) acc[r] = 0.0;
        vecs<0>(win, acc, taps, flip);
    }
};

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// POLYNOMIAL edge rows (reference savgolFilter.c:773-784) as items of the tile kernels: for each channel end n outputs, each its
// own 2n+1-tap row of edge_weights.  One wave per (channel, end); lanes hold the taps (lane l: taps l and l + 64), the dot
// product is a wavefront butterfly sum.
//   leading : out[e]       = s * sum_k ew[e][k] * in[2n - k]        (reversed data: the reference's odd-derivative sign quirk
//                                                                    included; JOB_EDGE_NEGATE is the opt-in fix)
//   trailing: out[L-1-e]   = s * sum_k ew[e][k] * in[L - ws + k]
// Rounds 1-2 ran this as a kernel of its own (same arithmetic, bit for bit): the n outputs of a channel end were n dependent
// round trips to the edge table -- 14 us at n = 32, a third of a single-signal call (POLYNOMIAL 33.9 us against REFLECT 20.2 for
// 10^6 samples, profiles/r03_host_time_per_call.txt) -- and a second launch.  Here the half window is a template parameter: all
// 2n table loads are issued up front, and the item runs beside the channel's tiles.
template <typename T, int N, typename Load, typename Store>
__device__ __forceinline__ void sg1d_edge_rows(const float *__restrict__ ew, unsigned flags, float dt_inv, bool trailing, long long L, int lane,
                                               Load sample, Store put)
{
    constexpr int WS = 2 * N + 1;
    const int k0 = lane, k1 = lane + 64;
    T x0 = T(0), x1 = T(0);
    if (k0 < WS) x0 = sample(trailing ? (L - WS + k0) : (long long)(2 * N - k0));
    if (WS > 64 && k1 < WS) x1 = sample(trailing ? (L - WS + k1) : (long long)(2 * N - k1));
    float w0[N], w1[N];
#pragma unroll
    for (int e = 0; e < N; ++e) {
        w0[e] = k0 < WS ? ew[e * WS + k0] : 0.0f;
        w1[e] = (WS > 64 && k1 < WS) ? ew[e * WS + k1] : 0.0f;
    }
    // products and the butterfly sum in double for fp32 data too (round 4): an edge row of a high-order fit has large taps of both signs
    // (n = 5, poly_order 6: the fp32 sum was 2 x the reference's own error, the worst case of the whole accuracy sweep), and these
    // 2n outputs per channel cost nothing
#pragma unroll
    for (int e = 0; e < N; ++e) {
        double p = 0.0;
        if (k0 < WS) p = (double)w0[e] * (double)x0;
        if (WS > 64 && k1 < WS) p = __builtin_fma((double)w1[e], (double)x1, p);
        p = wave_sum(p);
        if (flags & JOB_SCALE) p *= (double)dt_inv;
        if ((flags & JOB_EDGE_NEGATE) && !trailing) p = -p;
        if (lane == 0) put(trailing ? (L - 1 - e) : (long long)e, (T)p);
    }
}

// item 2c = leading end of channel c, 2c + 1 = its trailing end
template <typename T, int N>
__device__ __forceinline__ void sg1d_edge_item(const Job1D &job, unsigned item, int lane)
{
    const long long c = item >> 1;
    const T *__restrict__ row = static_cast<const T *>(job.in) + c * job.in_ld;
    T *__restrict__ orow = static_cast<T *>(job.out) + c * job.out_ld;
    const bool trailing = (item & 1u) != 0;
    if (job.edge_stash) {
        // in place: the 2n+1 samples of this channel end as they were before any tile stored (Job1D::edge_stash)
        const T *es = static_cast<const T *>(job.edge_stash) + (long long)item * (2 * N + 1);
        const long long base = trailing ? (long long)job.length - (2 * N + 1) : 0;
        sg1d_edge_rows<T, N>(job.edges, job.flags, job.dt_inv, trailing, (long long)job.length, lane,
                             [&](long long i) { return es[i - base]; }, [&](long long i, T v) { orow[i] = v; });
        return;
    }
    sg1d_edge_rows<T, N>(job.edges, job.flags, job.dt_inv, trailing, (long long)job.length, lane,
                         [&](long long i) { return row[i]; }, [&](long long i, T v) { orow[i] = v; });
}
// the same on array-of-structs data (always the edges of a strided call: the reference's savgol_apply_strided ignores config.boundary, :902-928)
template <int N>
__device__ __forceinline__ void sg1d_edge_item(const JobStrided &job, unsigned item, int lane)
{
    const long long c = item >> 1;
    const char *__restrict__ row = job.in + c * job.in_pitch;
    char *__restrict__ orow = job.out + c * job.out_pitch;
    sg1d_edge_rows<float, N>(job.edges, job.flags, job.dt_inv, (item & 1u) != 0, (long long)job.length, lane,
                             [&](long long i) { return *reinterpret_cast<const float *>(row + i * job.in_stride); },
                             [&](long long i, float v) { *reinterpret_cast<float *>(orow + i * job.out_stride) = v; });
}

// Work distribution: ONE TILE PER WAVE, blocks dispatched in order (grid = total_tiles / 4).  Round 1 ran a persistent
// grid (resident waves striding over the tiles, next tile prefetched into registers); measured on MI355X that shape caps a
// read+write stream at 5.2-5.5 TB/s, while the same tiles handed out by the hardware dispatcher in block order stream at
// 5.7-6.2 TB/s (tools/membench2.hip, tools/fmastream2.hip: 7.0 -> 6.5 ms for this kernel's bytes, FMAs and LDS traffic).
// Latency is hidden across the 4 waves a SIMD holds, not inside a wave.  Blocks that share an XCD (blockIdx % 8, observed
// round-robin placement) get neighbouring tiles -- each XCD sweeps its own eighth of the batch -- so the halo a tile shares
// with its neighbour is an L2 hit; placement affects speed only.  Channel-end tiles (slower, see below) are simply tiles
// that take longer; the dispatcher balances them.
template <typename T, int N, typename CV>
__device__ __forceinline__ void sg1d_tile_body(const Job1D &job, const typename CV::Args &taps)
{
    typedef typename CV::K K;                                // tile geometry of the convolution policy (narrow or wide tiles)
    typedef typename K::VT VT;
    constexpr int E = K::E, R = K::R, TW = K::TW, NA = K::NA, HV = K::HV, VPL = K::VPL, TV = K::TV;

    __shared__ __attribute__((aligned(16))) char smem[K::WAVES * K::SLAB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, lives in an SGPR
    char *slab = smem + wave * K::SLAB;

    const unsigned nb8 = gridDim.x >> 3;
    unsigned blk = blockIdx.x;
    if (blk < nb8 * 8u) {
        // (the host counts chunks in blocks of four tiles: a kernel family built with fewer waves per block has proportionally more blocks per chunk)
        const unsigned cs = job.xcd_chunk_log2 == 0 || job.xcd_chunk_log2 >= 32u ? job.xcd_chunk_log2 : job.xcd_chunk_log2 + (K::WAVES == 4 ? 0u : K::WAVES == 2 ? 1u : 2u);
        if (cs == 0) blk = (blk & 7u) * nb8 + (blk >> 3);
        else if (cs < 32u) {
            const unsigned span = 8u << cs, q = blk >> (cs + 3u);
            if ((q + 1u) * span <= nb8 * 8u) { const unsigned r = blk & (span - 1u); blk = (((q << 3) + (r & 7u)) << cs) + (r >> 3); }   // the last, partial span keeps launch order
        }
    }
    const unsigned tile = blk * K::WAVES + wave;
    if (tile >= job.total_tiles) {                                        // wave-uniform: past the tiles come the edge items, if any
        if (tile - job.total_tiles < job.edge_items) sg1d_edge_item<T, N>(job, tile - job.total_tiles, lane);
        return;
    }

    const T *__restrict__ gin  = static_cast<const T *>(job.in);
    T *__restrict__       gout = static_cast<T *>(job.out);
    const int L = (int)job.length;
    const int mode = (int)(job.flags & JOB_MODE_MASK);

#ifdef SG_STAMPS
    const bool stamp_on = (blockIdx.x == 8 && wave == 1);
    const int stamp_it = 0;
#endif
    SG_STAMP(0);
    const unsigned c = job.tpc_shift >= 32 ? tile : (__umulhi(tile, job.tpc_magic) >> job.tpc_shift);   // tile / tiles_per_channel, on the scalar unit
    // in-place colour phases (Job1D::phase): the launch holds the even (1) or the odd (2) tiles of every channel.  Stash layout (Job1D): one slot of
    // [left NA | right NA] per ODD tile (slot of tile k = c * (T / 2) + (k - 1) / 2), and per channel `ends` = [tile 0's left | the last tile's right |
    // tile T-2's right when the last tile is shorter than NA | pad], NA samples each
    unsigned k = tile - c * job.tiles_per_channel;
    bool st_left = false, st_right = false;
    const VT *stl = nullptr, *str = nullptr;                              // where a stashed left / right halo comes from
    unsigned odd_slots = 0;
    if (job.phase) {
        k = 2u * k + (job.phase - 1u);
        odd_slots = job.tpc_all >> 1;
        if (job.phase == 1u) {
            st_left = k == 0u;
            const int last_body = (int)job.length - (int)(job.tpc_all - 1u) * TW;          // samples in the channel's last tile
            const bool last = k + 1u == job.tpc_all;
            st_right = last || (k + 2u == job.tpc_all && last_body < NA);
            stl = static_cast<const VT *>(job.ends) + (size_t)c * (4 * HV);
            str = stl + (last ? HV : 2 * HV);
        } else {
            st_left = st_right = true;
            stl = static_cast<const VT *>(job.stash) + ((size_t)c * odd_slots + (k >> 1)) * (2 * HV);
            str = stl + HV;
        }
    }
    const int ts = (int)k * TW;
    const T *__restrict__ row = gin + (long long)c * job.in_ld;
    // slab byte offset of vector lane + 64*s: when VPL divides 64 the pad count splits, (lane + 64 s)/VPL = lane/VPL + s*64/VPL,
    // so one VGPR holds the lane part and s goes into the instruction's immediate offset
    char *const slab_row = slab + slab_vec_off<VPL>(lane);
    auto row_vec = [&](int s) -> VT * {
        if constexpr (64 % VPL == 0) return reinterpret_cast<VT *>(slab_row + s * (16 * (64 + 64 / VPL)));
        else return reinterpret_cast<VT *>(slab + slab_vec_off<VPL>(lane + 64 * s));
    };

    // ---- stage tile + halo into the slab ----
    if (st_left || st_right) {
        // IN PLACE (out == in): the body [ts, ts + TW) belongs to this tile alone -- nobody else reads or writes it.  A halo the neighbours may already
        // have overwritten (odd phase: both) or that reaches past the channel's end (even phase) comes from the tile's stash slot, already remapped /
        // zero-filled per mode; the other side of an even tile is still untouched in the rows
        if ((job.flags & JOB_VEC_IN) && ts + TW <= L) {
            const VT *src = reinterpret_cast<const VT *>(row + ts) - HV;              // slab vector v <-> sample ts - NA + v E
            VT p[VPL + 1];
#pragma unroll
            for (int s = 0; s < VPL; ++s) {
                if (s == 0) p[0] = (lane < HV && st_left) ? stl[lane] : ld_stream(src + lane);
                else p[s] = ld_stream(src + lane + 64 * s);
            }
            if (lane < 2 * HV) p[VPL] = (lane < HV || !st_right) ? ld_stream(src + TV + lane) : str[lane - HV];          // body tail, then the right halo
#pragma unroll
            for (int s = 0; s < VPL; ++s) *row_vec(s) = p[s];
            if (lane < 2 * HV) *row_vec(VPL) = p[VPL];
        } else {
            // the last tile of a channel (its body ends at L) and rows without 16-byte alignment: element by element
            const T *sel = reinterpret_cast<const T *>(stl), *ser = reinterpret_cast<const T *>(str);
            const int tend = ts + TW < L ? ts + TW : L;
#pragma unroll 4
            for (int e = lane; e < K::SL; e += 64) {
                const int g = ts - NA + e;
                T x = T(0);
                if (g < ts) x = st_left ? sel[e] : row[g];
                else if (g < tend) x = row[g];
                else if (g < tend + NA) x = st_right ? ser[g - tend] : row[g];
                *reinterpret_cast<T *>(slab + slab_vec_off<VPL>(e / E) + (e % E) * (int)sizeof(T)) = x;
            }
        }
    } else
    // a tile is "full" when every 16-B vector of tile + halo lies inside the row: the common case
    if ((job.flags & JOB_VEC_IN) && ts - NA >= 0 && ts + TW + NA <= L) {
        const VT *src = reinterpret_cast<const VT *>(row + (ts - NA));
        VT p[VPL + 1];
#pragma unroll
        for (int s = 0; s < VPL; ++s) p[s] = ld_stream(src + lane + 64 * s);
        if (lane < 2 * HV) p[VPL] = src[TV + lane];                       // halo: re-read by the neighbour tile, keep it cached
#pragma unroll
        for (int s = 0; s < VPL; ++s) *row_vec(s) = p[s];
        if (lane < 2 * HV) *row_vec(VPL) = p[VPL];
    } else {
        // Channel ends, short rows, rows without 16-B alignment.  Vectors that lie wholly inside the
        // row are still moved as vectors; the rest (the part of the halo that sticks out of the row,
        // remapped per boundary mode; everything if the row is unaligned) goes element by element.
        const bool vec = (job.flags & JOB_VEC_IN) != 0;
        const int lim = L + NA;                                      // nothing beyond is ever used
#pragma unroll
        for (int s = 0; s < VPL + 1; ++s) {
            const int v = lane + 64 * s;
            const int g0 = ts - NA + v * E;
            if (v < K::SV && vec && g0 >= 0 && g0 + E <= L)
                *reinterpret_cast<VT *>(slab + slab_vec_off<VPL>(v)) = *reinterpret_cast<const VT *>(row + g0);
        }
#pragma unroll 4
        for (int e = lane; e < K::SL; e += 64) {
            int g = ts - NA + e;
            const int g0 = g - (e % E);
            const bool direct = vec && g0 >= 0 && g0 + E <= L;
            if (!direct) {
                // beyond L + NA nothing a STORED output needs is read -- but the slab is whatever the previous block left there, and round 5's
                // x-stationary inner product multiplies the sample one past a window by a zero tap (0 x NaN): zero-fill instead of skipping
                T x = T(0);
                if (g < lim) {
                    bool zero = false;
                    if (g < 0 || g >= L) g = remap_index(g, L, mode, zero);
                    if (!zero) x = row[g];
                }
                *reinterpret_cast<T *>(slab + slab_vec_off<VPL>(e / E) + (e % E) * (int)sizeof(T)) = x;
            }
        }
    }
    SG_STAMP(1);
    wave_lds_sync();
    SG_STAMP(2);
    if (job.phase == 1u) {
        // even phase: the first NA input samples of this body are the RIGHT halo of tile k - 1, the last NA the LEFT halo of tile k + 1 -- into
        // their slots before this tile's results overwrite the rows (the slab holds them: vectors HV .. 2 HV - 1 and TV .. TV + HV - 1; positions
        // past the channel's end hold the remapped values the staging put there, which is what the neighbour's window needs)
        VT *slots = const_cast<VT *>(static_cast<const VT *>(job.stash)) + (size_t)c * odd_slots * (2 * HV);       // this channel's odd slots: tile 2j + 1 -> slot j
        if (lane < HV) {
            if (k > 0u) slots[(size_t)((k >> 1) - 1u) * (2 * HV) + HV + lane] = *reinterpret_cast<const VT *>(slab + slab_vec_off<VPL>(HV + lane));        // tile k - 1's right halo
        } else if (lane < 2 * HV) {
            if (k + 1u < job.tpc_all) slots[(size_t)(k >> 1) * (2 * HV) + (lane - HV)] = *reinterpret_cast<const VT *>(slab + slab_vec_off<VPL>(TV + lane - HV));   // tile k + 1's left halo
        }
    }

    // ---- derivative filters (JOB_CENTRE, fp32): centre the tile.  sum_k w_k x_k = sum_k w_k (x_k - c) + c sum_k w_k for any c; with c = the mean of the
    // tile's body the partial sums / block moments below meet the signal's variation across the tile instead of its offset.  A filter whose
    // weights sum to ~0 on a signal riding on a large offset stood at 1.1-1.4 x the reference's own error (block moments from half window 20: the
    // blocks' shares cancel only after each has been rounded at the offset's size; tools/offset_probe_1d.py, R6.16).  Smoothing filters do not
    // come here (the flag is off: nothing of this runs).  After the in-place slots have been written -- those must hold the raw samples.
    T centre = T(0);
    if (job.flags & JOB_CENTRE) {                          // uniform
        // c = the MEAN of the tile's body (a single sample of a zero-mean signal would double what the sums meet): every lane's partial sum, then an
        // xor butterfly -- both partners of a step add the same two numbers, so all 64 lanes end with the same bits
        VT p[VPL + 1];
        T part = T(0);
#pragma unroll
        for (int s = 0; s < VPL + 1; ++s) {
            if (s < VPL || lane < 2 * HV) {
                p[s] = *row_vec(s);
                // slab vectors HV .. HV + TV - 1 are the body: vector index lane + 64 s
                const int v = lane + 64 * s;
                if (v >= HV && v < HV + TV) {
#pragma unroll
                    for (int e = 0; e < E; ++e) part += vget(p[s], e);
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        centre = part * (T(1) / T(TW));
        if (!(centre - centre == T(0))) centre = T(0);     // Inf / NaN in the tile: leave it as it is
#pragma unroll
        for (int s = 0; s < VPL + 1; ++s) {
            if (s < VPL || lane < 2 * HV) {
#pragma unroll
                for (int e = 0; e < E; ++e) vset(p[s], e, vget(p[s], e) - centre);
                *row_vec(s) = p[s];
            }
        }
        wave_lds_sync();
    }

    // ---- the convolution: lane owns outputs [lane*R, lane*R + R) of the tile ----
    T acc[R];
    CV::run(slab + 16 * (lane * (VPL + 1)), taps, acc, job.flags);
    if (job.flags & JOB_CENTRE) {
        const T back = centre * (T)job.centre_sum;
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] += back;
    }
    if (job.flags & JOB_SCALE) {
        const T s = (T)job.dt_inv;
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] *= s;
    }
    wave_lds_sync();                                   // all window reads done before overwrite
    SG_STAMP(3);

    // ---- results back through the slab, then coalesced rows to HBM ----
    constexpr bool SWZ = (VPL == 8);                     // the swizzled result layout (result_vec_off8); other tile widths keep the padded one
    {
        // lane's vector s: byte 128 lane + 16 (s ^ lane % 8) = (128 lane + 16 (lane % 8)) ^ (16 s): one v_xor_b32 with a literal per write
        const int wbase = SWZ ? 128 * lane + 16 * (lane & 7) : 16 * (lane * (VPL + 1));
#pragma unroll
        for (int s = 0; s < VPL; ++s) {
            VT o;
#pragma unroll
            for (int e = 0; e < E; ++e) vset(o, e, acc[s * E + e]);
            *reinterpret_cast<VT *>(slab + (SWZ ? (wbase ^ (16 * s)) : wbase + 16 * s)) = o;
        }
    }
    wave_lds_sync();
    T *__restrict__ orow = gout + (long long)c * job.out_ld - (long long)job.out_shift;
    const int lo = (int)job.store_lo, hi = (int)job.store_hi;
    const bool whole = (job.flags & JOB_VEC_OUT) && ts >= lo && ts + TW <= hi;
    if (whole) {
        // vector lane + 64 s of the tile: with the swizzle its offset splits into a lane part (one VGPR) and 1024 s (an immediate)
        const char *const rbase = slab + (SWZ ? result_vec_off8(lane) : 0);
#pragma unroll
        for (int s = 0; s < VPL; ++s) {
            const VT o = SWZ ? *reinterpret_cast<const VT *>(rbase + 1024 * s) : *row_vec(s);
            st_stream(reinterpret_cast<VT *>(orow + ts) + lane + 64 * s, o);
        }
    } else {
        // first / last tile of a channel (the stored range ends inside it) or unaligned output rows
        const bool vec = (job.flags & JOB_VEC_OUT) != 0;
#pragma unroll
        for (int s = 0; s < VPL; ++s) {
            const int p = lane + 64 * s;
            const int g0 = ts + p * E;
            const VT o = *reinterpret_cast<const VT *>(slab + (SWZ ? result_vec_off8(p) : slab_vec_off<VPL>(p)));
            if (vec && g0 >= lo && g0 + E <= hi) {
                *reinterpret_cast<VT *>(orow + g0) = o;
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (g0 + e >= lo && g0 + e < hi) orow[g0 + e] = vget(o, e);
            }
        }
    }
    SG_STAMP(4);
}

// the plain sliding dot product: every output is 2N+1 multiply-adds (Conv<T, N> above)
template <typename T, int N, int V>
struct DirectConv {
    typedef Taps Args;
    typedef K1D<T, N, V> K;
    static __device__ __forceinline__ void run(const char *win, const Taps &taps, T (&acc)[K::R], unsigned flags)
    {
        if constexpr (sizeof(T) == 8) Conv<T, N, V>::run(win, taps, acc, (flags & JOB_ODD_TAPS) ? 0x80000000u : 0u);
        else Conv<T, N, V>::run(win, taps, acc);
    }
};

// V = vectors per lane: the narrow tile every job can use, or the wide one the host picks for big batches (sg_k1d_host.hpp)
template <typename T, int N, int V>
__global__ __launch_bounds__(64 * SG_K1D_WAVES, (K1D<T, N, V>::MIN_WAVES)) void sg1d_center_kernel(const Job1D job, const Taps taps)
{
    sg1d_tile_body<T, N, DirectConv<T, N, V>>(job, taps);
}

// ---------------------------------------------------------------------------------------------
// The same tile on array-of-structs data (reference savgol_apply_strided, savgolFilter.c:877-934: every window is copied out
// of the records into a 65-float stack buffer, :904-906).  Here the FIELD is gathered while the tile is staged -- lane l takes
// samples l, l + 64, ... of tile + halo, one dword load each, straight into the slab -- and scattered from the slab when the
// results leave: one pass over the records instead of round 2's gather kernel -> dense kernels -> scatter kernel (three passes
// and two dense scratch frames).  Everything between staging and store is the dense kernel's (Conv<float, N>, same slab).
// ---------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(64 * SG_K1D_WAVES, (K1D<float, N, SG_VPL_NARROW>::MIN_WAVES)) void sg1d_strided_kernel(const JobStrided job, const Taps taps)
{
    typedef K1D<float, N, SG_VPL_NARROW> K;
    constexpr int R = K::R, TW = K::TW, NA = K::NA, VPL = K::VPL;
    __shared__ __attribute__((aligned(16))) char smem[K::WAVES * K::SLAB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *slab = smem + wave * K::SLAB;

    const unsigned nb8 = gridDim.x >> 3;
    unsigned blk = blockIdx.x;
    if (blk < nb8 * 8u) blk = (blk & 7u) * nb8 + (blk >> 3);
    const unsigned tile = blk * K::WAVES + wave;
    if (tile >= job.total_tiles) {                                        // wave-uniform: past the tiles come the edge items
        if (tile - job.total_tiles < job.edge_items) sg1d_edge_item<N>(job, tile - job.total_tiles, lane);
        return;
    }
    const int L = (int)job.length;
    const int mode = (int)(job.flags & JOB_MODE_MASK);
    const unsigned c = job.tpc_shift >= 32 ? tile : (__umulhi(tile, job.tpc_magic) >> job.tpc_shift);
    const int ts = (int)(tile - c * job.tiles_per_channel) * TW;
    const char *__restrict__ row = job.in + (long long)c * job.in_pitch;
    char *__restrict__ orow = job.out + (long long)c * job.out_pitch;

    // slab byte offset of element i = lane + 64 k: vector i/4 = lane/4 + 16 k, and with VPL = 8 the pad count splits the same way
    static_assert(VPL == 8, "the element <-> slab offset split below is written for 8 vectors per lane");
    char *const mine = slab + 16 * ((lane >> 2) + (lane >> 5)) + 4 * (lane & 3);
    constexpr int KSTEP = 16 * 18;                                        // 16 vectors + 2 pads per 64 elements
    constexpr int NK = (K::SL + 63) / 64;

    // ---- stage tile + halo: sample g = ts - NA + i of this channel's field ----
    if (ts - NA >= 0 && ts + TW + NA <= L) {
        const char *p = row + (long long)(ts - NA + lane) * job.in_stride;
        float x[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k)
            if (lane + 64 * k < K::SL) x[k] = *reinterpret_cast<const float *>(p + (long long)(64 * k) * job.in_stride);
#pragma unroll
        for (int k = 0; k < NK; ++k)
            if (lane + 64 * k < K::SL) *reinterpret_cast<float *>(mine + k * KSTEP) = x[k];
    } else {
        // channel ends: samples outside the row through the boundary mode's remap (get_padded_sample, :442-482); POLYNOMIAL
        // (and anything unknown) reads zeros there -- the outputs that would see them are the edge kernel's, not stored here
#pragma unroll 4
        for (int k = 0; k < NK; ++k) {
            const int i = lane + 64 * k;
            int g = ts - NA + i;
            if (i < K::SL && g < L + NA) {
                bool zero = false;
                if (g < 0 || g >= L) g = remap_index(g, L, mode, zero);
                const float x = zero ? 0.0f : *reinterpret_cast<const float *>(row + (long long)g * job.in_stride);
                *reinterpret_cast<float *>(mine + k * KSTEP) = x;
            }
        }
    }
    wave_lds_sync();

    float acc[R];
    Conv<float, N, SG_VPL_NARROW>::run(slab + 16 * (lane * (VPL + 1)), taps, acc);
    if (job.flags & JOB_SCALE) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] *= job.dt_inv;
    }
    wave_lds_sync();                                   // all window reads done before overwrite
#pragma unroll
    for (int s = 0; s < VPL; ++s)
        *reinterpret_cast<float4 *>(slab + 16 * (lane * (VPL + 1) + s)) = float4{acc[4 * s], acc[4 * s + 1], acc[4 * s + 2], acc[4 * s + 3]};
    wave_lds_sync();

    // ---- results out of the slab, element lane + 64 k of the tile -> its record ----
    const int lo = (int)job.store_lo, hi = (int)job.store_hi;
    char *q = orow + (long long)(ts + lane) * job.out_stride;
    // (Round 4 tried whole-record stores for two fields of the SAME 8- / 16-byte records -- re-read the record, replace the field, store
    //  8 / 16 bytes per lane: 265 / 145 Gsamples/s against 324 / 159 with the plain 4-byte stores below.  A partial store into a line the
    //  tile has just read merges in L2; only a destination array of its own pays the memory a read-modify-write: profiles/r04_strided.txt.)
    if (ts >= lo && ts + TW <= hi) {
#pragma unroll
        for (int k = 0; k < TW / 64; ++k)
            *reinterpret_cast<float *>(q + (long long)(64 * k) * job.out_stride) = *reinterpret_cast<const float *>(mine + k * KSTEP);
    } else {
#pragma unroll 4
        for (int k = 0; k < TW / 64; ++k) {
            const int g = ts + lane + 64 * k;
            if (g >= lo && g < hi) *reinterpret_cast<float *>(q + (long long)(64 * k) * job.out_stride) = *reinterpret_cast<const float *>(mine + k * KSTEP);
        }
    }
}

}  // namespace sg
