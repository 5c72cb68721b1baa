// sg_stream_dma.hip -- savgol_streambank_push_block, LDS-DMA tile form (round 5).
//
// Reference loop: src/savgol_stream.c:25-38 (the ring dot product) driven by :152-178 (one push per sample) -- here `ticks` pushes of
// every stream in one launch, a convolution down the tick axis of samples[tick][stream].
//
// Why another form.  The walk (sg_bank_roll_kernel) keeps few rows in flight per wave; the register tiles (sg_bank_tile_kernel) load
// TR + 2n rows up front but hold them ALL in VGPRs until the last output is done (212 registers at n = 16: two waves per SIMD), and a
// wave that computes issues no memory traffic -- bare, that access pattern streams at 0.72-0.73 of 8 TB/s, with the arithmetic in it at
// 0.60-0.66 (tools/membench_streamtile.hip, profiles/r05_membench_streamtile.txt).  Here the rows never touch a VGPR on their way in:
//   * a wave owns 128 adjacent streams (a lane 2 = one 8-byte LDS read per row) x TR ticks and a private LDS slab of TR + 2n rows;
//   * it issues every row load up front as LDS-DMA (global_load_lds_dwordx4: 1 KiB = two 512-byte rows per instruction, no VGPR
//     destination), then consumes the rows IN ARRIVAL ORDER behind counted s_waitcnt vmcnt(k): each arriving row is read once from the
//     slab and fed into every accumulator it touches ("input stationary": row r is tap r - m of output m).  The arithmetic overlaps the
//     arrival of the later rows inside ONE wave, registers hold only the TR accumulators, and a finished output row is stored at once;
//   * per output the arithmetic is bank_roll_item's, bit for bit in both banks: reference order = one accumulator from 0, taps ascending,
//     multiply and add rounded separately (rows arrive in ascending tap order for every output); FMA = two chains (even / odd taps);
//   * tiles are dealt like the register tiles': blocks that share an XCD walk down a group of neighbouring strips band by band, so the 2n
//     halo rows a tile shares with the one above are L2 hits.
// Bare, this pattern streams at 0.75-0.76 (the flat copy on the same box: 0.79).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "sg_internal.h"
#include "sg_pk.hpp"
#include "sg_runtime.hpp"
#include "sg_stream_roll.hpp"

namespace sg {

#ifndef SG_DMA_WPB
#define SG_DMA_WPB 4                                         // waves per block (independent: only how waves are dealt to CUs and how LDS is carved)
#endif
#ifndef SG_DMA_STORE_AUX
#define SG_DMA_STORE_AUX 2                                   // cache policy bits of the output stores: 2 = nontemporal (written once, never read by this launch:
                                                             // 0.392 against 0.412 ms on one box, profiles/r05_stream_dma.txt; nontemporal row LOADS lose 30 %: the halo rows are re-read)
#endif
#ifndef SG_DMA_LOAD_NT
#define SG_DMA_LOAD_NT 0
#endif
#ifndef SG_DMA_MAX_N
#define SG_DMA_MAX_N 32
#endif

// output ticks per tile TR: the slab is TR + 2n rows of 512 bytes, the accumulators 2 (reference order) or 4 (two FMA chains) VGPRs per tick
template <int N, int TR_> struct DmaShape {
    static constexpr int TR = TR_;
    static constexpr int ROWS = TR + 2 * N, NI = ROWS / 2, RB = 512, SLAB = ROWS * RB;
    static_assert(TR % 2 == 0, "a DMA instruction moves two rows");
};
// defaults (A/B'd on config 3's shape, profiles/r05_stream_dma.txt)
#ifndef SG_DMA_TR
#define SG_DMA_TR 32
#endif
#ifndef SG_DMA_PAIRS
#define SG_DMA_PAIRS 12                                      // row pairs (KiB) of LDS ring per wave = how far the row loads run ahead of the arithmetic
                                                             // (8 / 12 / 16 / 24: FMA bank 0.401 / 0.401 / 0.405 / 0.479 ms, bit-exact bank sustained 0.551 / 0.545-0.555 /
                                                             //  0.579-0.592 / 0.710 on one box: a shallow ring leaves LDS for the waves the bit-exact arithmetic needs)
#endif

template <int K> __device__ __forceinline__ void wait_vm()
{
    static_assert(K >= 0 && K < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K) : "memory");
}

// one 1 KiB LDS-DMA: lane l's 16 bytes land at lds_dst + 16 l.  Inline asm (the compiler then neither counts it nor drains it at the
// first LDS read: the waits are counted by hand below); M0 is the compiler's, so it is saved and restored in the same statement.
__device__ __forceinline__ void dma16(const float *gsrc, unsigned lds_dst)
{
    unsigned keep;
#if SG_DMA_LOAD_NT
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
#else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
#endif
}

// vmcnt bookkeeping, all at compile time.  The wave's vector-memory queue, in issue order: the DP DMAs of the prologue, then per step j
// (= row pair j): the stores of the outputs that rows 2j, 2j + 1 finish, then DMA j + DP (if that pair exists).
template <int N, int TR, int DP> struct DmaQueue {
    static constexpr int NI = (TR + 2 * N) / 2;
    static constexpr int done(int r) { return r - 2 * N < 0 ? 0 : (r - 2 * N > TR ? TR : r - 2 * N); }     // outputs finished by rows < r = stores issued
    static constexpr int dmas(int a, int b) { int c = 0; for (int j = a; j < b; ++j) c += (j >= 0 && j + DP < NI) ? 1 : 0; return c; }  // DMAs issued by steps [a, b)
    // operations younger than DMA p when step g starts (p = g + 1 is the pair step g waits for)
    static constexpr int younger(int p, int g)
    {
        if (p < DP) return (DP - 1 - p) + done(2 * g) + dmas(0, g);                     // a prologue DMA: the rest of the prologue, then every step so far
        const int born = p - DP;                                                        // issued at the END of step `born`
        return (done(2 * g) - done(2 * born + 2)) + dmas(born + 1, g);
    }
};

// ---- block moments down the tick axis (round 5, profiles/EXPERIMENTS.md R5.9) ----
// The fused bank's taps are a polynomial in the tap index for the filters streams are made of (config 3: m = 2, d = 1 -- LINEAR taps).  A tile's rows
// are cut into blocks of 8 ticks (tile-relative rows 8j .. 8j + 7); output m = 8a + p takes the rows of its window that fill whole blocks through the
// blocks' M moments (M coefficients per block, the same for both streams of the lane) and only the rows before its first / after its last whole block
// tap by tap: at n = 16, 7 direct taps + 3.25 blocks x M per output on average, plus M operations per input row for the moments -- 17.5 (M = 2)
// instead of 33 packed multiply-adds per output pair.  Same idea as sg_k1d_momenth.hpp, along the other axis.
template <int N> struct MomGeom {
    static constexpr int BK = 8;
    static constexpr int NOFF = 2 * N - (BK - 1) + 1;                                    // block offsets 8j - m = 0 .. 2N - 7
    static constexpr int jf(int m) { return m / BK + (m % BK ? 1 : 0); }                 // first / last block that lies wholly inside the window m .. m + 2N
    static constexpr int jl(int m) { return (m + 2 * N - (BK - 1)) / BK; }
    static constexpr bool whole(int m, int r) { return r >= m && r <= m + 2 * N && r / BK >= jf(m) && r / BK <= jl(m); }
    static constexpr bool direct(int m, int r) { return r >= m && r <= m + 2 * N && !whole(m, r); }
    static_assert(N >= BK / 2, "a window holds at least one whole block");
};
// q_0 = 1, q_1(t) = t - 3.5, q_2(t) = (t - 3.5)^2 - 5.25 on t = 0..7 (orthogonal; every value exact in fp32)
template <int N, int M> struct MomTaps {
    f32x2 head[4];                                   // w[0 .. 7]        (direct taps before the first whole block: k <= 6)
    f32x2 tail[4];                                   // w[2N - 7 .. 2N]  (direct taps after the last whole block: k >= 2N - 6)
    f32x2 c[M][(MomGeom<N>::NOFF + 1) / 2];          // c[s][off]: the block at offset off = 8j - m contributes sum_s c[s][off] * moment_s
    f32x2 q[M > 1 ? M - 1 : 1][4];                   // q_s(t), s = 1 .. M - 1
};

template <int N, bool FMA, int TRT, int WPB, int DP, int FCH = 2, int MOM = 0, class TAPS = SRollTaps<N>>
__global__ __launch_bounds__(64 * WPB) void sg_bank_dma_kernel(const BankJob job, const TAPS taps, const TileGeom geo)
{
    typedef SRoll<N> R;
    typedef DmaShape<N, TRT> D;
    typedef DmaQueue<N, TRT, DP> Q;
    constexpr int TR = D::TR, NI = D::NI, RB = D::RB, RING = DP * 1024;
    static_assert(DP >= 2 && DP <= NI, "ring of row pairs");
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nblk = gridDim.x;
    const unsigned blk = (job.aligned & 2) ? blockIdx.x : (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3);
    const unsigned t = blk * WPB + (unsigned)wv;                                          // the launch keeps the tile count below 2^31
    if (t >= (unsigned)geo.total) return;
    // tile order: groups of `group` neighbouring strips; inside a group band after band, strips fastest (32-bit scalar divisions)
    const unsigned per_group = geo.group * geo.bands;
    const unsigned grp = t / per_group;
    const unsigned rem = t - grp * per_group;
    const unsigned gs = geo.strips - grp * geo.group < geo.group ? geo.strips - grp * geo.group : geo.group;
    const unsigned band = rem / gs, strip = grp * geo.group + (rem - band * gs);
    if (band >= geo.bands) return;
    const long long t0 = (long long)band * TR;
    const unsigned ring = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)lds) + (unsigned)wv * (unsigned)RING;   // LDS byte address, wave-uniform
    const int sub = lane >> 5, chunk = lane & 31;                                                 // row of the pair, 16-byte chunk of the row
    const size_t col = (size_t)strip * 128 + (size_t)chunk * 4;

    f32x2 cen = f32x2{0.0f, 0.0f}, backdt = f32x2{0.0f, 0.0f};       // fused bank, derivative filters: what the tile's rows are centred on (set before the first row is fed)
    // source of row pair i: slab row r = history index t0 - 2N + r (history: ring contents, then this call's samples)
    const bool inside = t0 >= 2 * N && t0 + TR <= (long long)job.ticks;                           // uniform: every row is one of this call's samples
    const float *const p0 = job.samples + (size_t)(inside ? t0 - 2 * N + sub : 0) * job.streams + col;
    const size_t pstep = 2 * job.streams;
    auto issue = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const float *src;
        if (inside) {
            src = p0 + (size_t)i * pstep;
        } else {
            long long h = t0 - 2 * N + 2 * i + sub;
            if (h >= (long long)job.ticks) h = (long long)job.ticks - 1;                          // past the call: loaded, never used
            int slot = job.wp0 + (int)(h < 0 ? h : 0);
            slot = slot < 0 ? slot + R::WS : slot;
            src = (h >= 0 ? job.samples + (size_t)h * job.streams : job.ring + (size_t)slot * job.streams) + col;
        }
        dma16(src, ring + (unsigned)(i % DP) * 1024u);
    };
    static_for<DP>([&](auto ic) -> bool { issue(ic); return true; });

    // ---- consume the rows in arrival order ----
    const char *mine = lds + (size_t)wv * RING + lane * 8;
    const int row_bytes = (int)(job.streams * 4);
    const unsigned voff = (strip * 128u + 2u * (unsigned)lane) * 4u;                              // byte offset of this lane's streams in a row
    constexpr int CH = MOM ? 1 : (FMA ? FCH : 1);                                                 // FMA bank: two chains (even / odd taps) or one; block moments: one (an output is ~14 terms)
    f32x2 acc[CH][TR];
    f32x2 mom[MOM > 0 ? MOM : 1];                                                                 // block moments of the block being filled (MOM > 0)
    auto row_in = [&](auto rc) -> f32x2 {
        constexpr int r = decltype(rc)::value;
        return *reinterpret_cast<const f32x2 *>(mine + ((r / 2) % DP) * 1024 + (r & 1) * RB);
    };
    auto feed = [&](auto rc, const f32x2 x) {
        constexpr int r = decltype(rc)::value;
        constexpr int mlo = r - 2 * N > 0 ? r - 2 * N : 0, mhi = r < TR - 1 ? r : TR - 1;
        if constexpr (MOM > 0) {
            typedef MomGeom<N> G;
            static_assert(N >= 8, "head taps (k <= 6) and tail taps (k >= 2N - 6) must not meet");
            constexpr int j = r / G::BK, t = r % G::BK;
            // the block's moments (tap-free: shared by every output that takes this block whole)
            if constexpr (t == 0) {
                mom[0] = x;
                static_for<MOM - 1>([&](auto sc) -> bool { constexpr int sm = decltype(sc)::value; mom[sm + 1] = pk_mul_sgpr<(t & 1)>(taps.q[sm][t >> 1], x); return true; });
            } else {
                mom[0] = mom[0] + x;
                static_for<MOM - 1>([&](auto sc) -> bool { constexpr int sm = decltype(sc)::value; pk_fma_sgpr<(t & 1)>(mom[sm + 1], taps.q[sm][t >> 1], x); return true; });
            }
            // rows before an output's first / after its last whole block: tap by tap
            static_for<mhi - mlo + 1>([&](auto ic) -> bool {
                constexpr int m = mlo + decltype(ic)::value, k = r - m;
                if constexpr (G::direct(m, r)) {
                    constexpr bool is_head = k < G::BK;
                    constexpr int kk = is_head ? k : k - (2 * N - (G::BK - 1));
                    static_assert(kk >= 0 && kk < 8, "direct taps sit within 7 of either end of the window");
                    if constexpr (k == 0) acc[0][m] = pk_mul_sgpr<(kk & 1)>(taps.head[kk >> 1], x);                    // m % 8 != 0: the output's first term
                    else if constexpr (is_head) pk_fma_sgpr<(kk & 1)>(acc[0][m], taps.head[kk >> 1], x);
                    else pk_fma_sgpr<(kk & 1)>(acc[0][m], taps.tail[kk >> 1], x);
                }
                return true;
            });
            // a block is complete: its share of every output that takes it whole
            if constexpr (t == G::BK - 1) {
                static_for<mhi - mlo + 1>([&](auto ic) -> bool {
                    constexpr int m = mlo + decltype(ic)::value;
                    if constexpr (G::whole(m, r)) {
                        constexpr int off = G::BK * j - m;
                        static_assert(off >= 0 && off < G::NOFF, "block offset");
                        static_for<MOM>([&](auto sc) -> bool {
                            constexpr int sm = decltype(sc)::value;
                            if constexpr (sm == 0 && off == 0) acc[0][m] = pk_mul_sgpr<(off & 1)>(taps.c[0][off >> 1], mom[0]);  // m % 8 == 0: the output's first term
                            else pk_fma_sgpr<(off & 1)>(acc[0][m], taps.c[sm][off >> 1], mom[sm]);
                            return true;
                        });
                    }
                    return true;
                });
            }
        } else if constexpr (FMA) {
            static_for<mhi - mlo + 1>([&](auto ic) -> bool {
                constexpr int m = mlo + decltype(ic)::value, k = r - m;
                // two chains (even taps, odd taps), one v_pk_fma_f32 per tap: bank_roll_item's fast form, bit for bit -- or ONE chain in the
                // reference's order (taller tiles fit the registers; each term rounds once where the reference rounds twice)
#ifdef SG_DMA_SKIP_TAPS           // timing experiment only (wrong results): how much of a tile's time the multiply-adds are (profiles/EXPERIMENTS.md R5.8)
                if constexpr (k >= CH && k >= N - SG_DMA_SKIP_TAPS && k <= N + SG_DMA_SKIP_TAPS) return true;
#endif
                if constexpr (k < CH) acc[k][m] = pk_mul_sgpr<k>(taps.w[0], x);
                else pk_fma_sgpr<(k & 1)>(acc[(k & 1) % CH][m], taps.w[k >> 1], x);
                return true;
            });
        } else {
            // the reference's order (src/savgol_stream.c:25-38): sum = 0; sum += w[k] * x[k], k ascending, product and sum rounded separately.
            // Volatile asm for products and sums alike, each product issued one output ahead of its sum: left to the compiler, all the
            // products of a row are hoisted in front of the sums and stay live (256 registers and scratch; see bank_accroll_item)
            f32x2 p;
            {
                constexpr int k0 = r - mlo;
                if constexpr ((k0 & 1) == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "s"(taps.w[k0 >> 1]), "v"(x));
                else                         asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(p) : "s"(taps.w[k0 >> 1]), "v"(x));
            }
            static_for<mhi - mlo + 1>([&](auto ic) -> bool {
                constexpr int m = mlo + decltype(ic)::value, k = r - m;
                f32x2 pn = p;
                if constexpr (m < mhi) {
                    constexpr int kn = k - 1;
                    if constexpr ((kn & 1) == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(pn) : "s"(taps.w[kn >> 1]), "v"(x));
                    else                         asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(pn) : "s"(taps.w[kn >> 1]), "v"(x));
                }
                if constexpr (k == 0) asm volatile("v_pk_add_f32 %0, %1, 0 op_sel_hi:[1,0]" : "=v"(acc[0][m]) : "v"(p));       // 0 + p: a product of -0 sums to +0, as in the reference
                else                  asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[0][m]) : "v"(p));
                p = pn;
                return true;
            });
        }
        if constexpr (r >= 2 * N && r - 2 * N < TR) {                                             // output m = r - 2N has seen its last row
            constexpr int m = r - 2 * N;
            f32x2 a = acc[0][m];
            if constexpr (CH == 2) a = a + acc[1][m];

            const long long tt = t0 + m;
            const bool has_out = tt < (long long)job.ticks && job.received0 + (unsigned long long)tt + 1 >= (unsigned long long)R::WS;   // uniform (reference :166-170)
            // fused bank: (a + c * sum_k w_k) * dt_inv in one multiply-add (backdt = c * sum_k w_k * dt_inv, zero unless the tile is centred)
            const f32x2 y = (MOM > 0 || FMA) ? __builtin_elementwise_fma(a, f32x2{job.dt_inv, job.dt_inv}, backdt) : a * f32x2{job.dt_inv, job.dt_inv};
            float *orow = job.out + (size_t)(tt < (long long)job.ticks ? tt : 0) * job.streams;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(orow, 0, has_out ? row_bytes : 0, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, y), rs, (int)voff, 0, SG_DMA_STORE_AUX);
        }
    };
    // Step g consumes pair g (already in xa, xb), after it has waited for pair g + 1 and issued its LDS reads (their latency hides behind
    // the arithmetic), and ends by issuing DMA g + DP into the ring slot pair g has just left.  A row without an output still issues its
    // store (into an empty descriptor), so the queue arithmetic above is static.
    // ---- fused bank, derivative filters (job.centre): c = the mean of EIGHT real samples of each stream (a single sample of a zero-mean stream would
    // double what the sums meet).  A tile whose rows are all real takes its first eight rows out of LDS once their four DMA pairs have landed -- no
    // memory request of its own: eight extra 8-byte loads per lane and tile cost config 3 12 % (the kernel lives on its request rate); the few tiles of
    // a bank's first 2N ticks (rows before the first sample: the ring's zeros, feeding only outputs that are not stored) load eight samples spread
    // from the oldest real one to the tile's last row.
    if constexpr (MOM > 0 || FMA) {
        static_assert(DP >= 4 && NI >= 4, "the first four row pairs are in the ring together");
        if (job.centre) {                                    // uniform; smoothing filters keep cen = 0
            f32x2 sum = f32x2{0.0f, 0.0f};
            if (t0 - 2 * N >= -(long long)job.received0) {   // uniform: row 0 is a real sample, and so are the rows behind it
                wait_vm<(Q::younger(3, 0) > 63 ? 63 : Q::younger(3, 0))>();
                static_for<8>([&](auto rc) -> bool { sum = sum + row_in(rc); return true; });
            } else {
                const long long h0 = -(long long)job.received0;
                long long h1 = t0 + TR - 1;
                if (h1 > (long long)job.ticks - 1) h1 = (long long)job.ticks - 1;
                const int span = (int)(h1 - h0);             // <= 2N + TR - 1
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const long long h = h0 + (span * i) / 7;
                    const float *src;
                    if (h >= 0) src = job.samples + (size_t)h * job.streams;
                    else { int slot = job.wp0 + (int)h; slot = slot < 0 ? slot + R::WS : slot; src = job.ring + (size_t)slot * job.streams; }
                    sum = sum + *reinterpret_cast<const f32x2 *>(src + (size_t)strip * 128 + 2 * (size_t)lane);
                }
            }
            cen = sum * f32x2{0.125f, 0.125f};
            if (!(cen.x - cen.x == 0.0f)) cen.x = 0.0f;           // an Inf / NaN among the eight: that stream's tile stays as it is
            if (!(cen.y - cen.y == 0.0f)) cen.y = 0.0f;
            backdt = cen * f32x2{job.centre_sum * job.dt_inv, job.centre_sum * job.dt_inv};
        }
    }
    wait_vm<(Q::younger(0, 0) > 63 ? 63 : Q::younger(0, 0))>();           // (a smaller count only waits longer: the counter has 6 bits)
    f32x2 xa = row_in(std::integral_constant<int, 0>{}), xb = row_in(std::integral_constant<int, 1>{});
    // The fused bank's tiles -- block moments and tap-by-tap alike, never the bit-exact bank -- run on CENTRED samples (end of round 6, R6.16): sum_k w_k x_k = sum_k w_k (x_k - c) + c sum_k w_k for any c; with c = the stream's own
    // first row of the tile the moments and their block offsets meet the signal's variation over 2N + TR ticks instead of its offset -- a derivative
    // filter (weights summing to ~0: config 3) on a stream riding on a large offset stood at 2.3 x the reference's own error, the blocks' shares
    // cancelling only after each had been rounded at the offset's size.  sum_k w_k comes from the reference's table (taps.sig).
    static_for<NI>([&](auto gc) -> bool {
        constexpr int g = decltype(gc)::value;
        f32x2 na = xa, nb = xb;
        if constexpr (g + 1 < NI) {
            wait_vm<(Q::younger(g + 1, g) > 63 ? 63 : Q::younger(g + 1, g))>();
            na = row_in(std::integral_constant<int, 2 * g + 2>{});
            nb = row_in(std::integral_constant<int, 2 * g + 3>{});
            __builtin_amdgcn_sched_barrier(0);                                                    // keep these reads AHEAD of pair g's arithmetic (left alone, hipcc sinks them to their use)
        }
        if constexpr (MOM > 0 || FMA) {
            feed(std::integral_constant<int, 2 * g>{}, xa - cen);
            feed(std::integral_constant<int, 2 * g + 1>{}, xb - cen);
        } else {
            feed(std::integral_constant<int, 2 * g>{}, xa);
            feed(std::integral_constant<int, 2 * g + 1>{}, xb);
        }
        if constexpr (g + DP < NI) {
            // pair g's slot is free: its two LDS reads were issued a step ago and their data has just been consumed -- but "consumed" is the
            // compiler's business, so drain the LDS queue explicitly before the DMA may overwrite the slot
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue(std::integral_constant<int, g + DP>{});
        }
        xa = na; xb = nb;
        return true;
    });
}

template <int N, bool FMA, int TRT, int WPB, int DPR, int FCH = 2>
[[maybe_unused]] static int launch_bank_dma(const float *center, BankJob job, hipStream_t st)
{
    typedef SRoll<N> R;
    typedef DmaShape<N, TRT> D;
    SRollTaps<N> taps;
    memset(&taps, 0, sizeof(taps));
    for (int k = 0; k < R::WS; ++k) {
        if (k & 1) taps.w[k >> 1].y = center[k]; else taps.w[k >> 1].x = center[k];
    }
    TileGeom geo;
    geo.strips = (unsigned)(job.streams / 128);
    geo.bands = (unsigned)((job.ticks + D::TR - 1) / D::TR);
    geo.group = 128u;                                                                            // 64 KiB of a tick row
    if (geo.group > geo.strips) geo.group = geo.strips;
    const unsigned groups = (geo.strips + geo.group - 1) / geo.group;
    geo.total = (unsigned long long)groups * geo.group * geo.bands;
    const unsigned long long blocks = (geo.total + WPB - 1) / WPB;
    if (geo.total >= 0x7fffff00ull) return 1;
    const unsigned grid = ((unsigned)blocks + 7u) & ~7u;
    job.aligned = 1;
    constexpr int DP = DPR < D::NI ? DPR : D::NI;                                                 // ring of row pairs (all of the tile's pairs: every load up front)
    constexpr size_t lds = (size_t)WPB * DP * 1024;
    static_assert(lds <= 160 * 1024, "slabs of one block must fit the CU's LDS");
    // more than 64 KiB of dynamic LDS needs the attribute, and the attribute belongs to the (function, DEVICE) pair: set per launch -- a process-wide
    // static applied it on whichever device launched first only (a block push is >= 0.1 ms, the call is host-side bookkeeping)
    if (lds > 65536 && hipFuncSetAttribute(reinterpret_cast<const void *>(sg_bank_dma_kernel<N, FMA, TRT, WPB, DP, FCH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return 1;
    }
    hipLaunchKernelGGL((sg_bank_dma_kernel<N, FMA, TRT, WPB, DP, FCH>), dim3(grid), dim3(64 * WPB), lds, st, job, taps, geo);
    return hipGetLastError() == hipSuccess ? 0 : 1;                                              // a refused launch = not covered: the caller falls back to the walk
}

#ifndef SG_DMA_MOM_BUILD
template <int N, bool FMA>
static int launch_bank_dma_shape(const float *center, const BankJob &job, hipStream_t st)
{
#ifdef SG_DMA_EXPERIMENT        // A/B build: tile height and waves per block picked per process (tools/experiments.sh stream_dma)
    static const int tr = [] { const char *e = getenv("SAVGOL_HIP_STREAM_DMA_TR"); return e ? atoi(e) : SG_DMA_TR; }();
    static const int wpb = [] { const char *e = getenv("SAVGOL_HIP_STREAM_DMA_WPB"); return e ? atoi(e) : SG_DMA_WPB; }();
    static const int dp = [] { const char *e = getenv("SAVGOL_HIP_STREAM_DMA_PAIRS"); return e ? atoi(e) : SG_DMA_PAIRS; }();
    static const int fch = [] { const char *e = getenv("SAVGOL_HIP_STREAM_DMA_CHAINS"); return e ? atoi(e) : 2; }();
#define SG_DMA_TRY(T, W, P) if (tr == T && wpb == W && dp == P && fch == 2) return launch_bank_dma<N, FMA, T, W, P, 2>(center, job, st);
#define SG_DMA_TRY1(T, W, P) if (tr == T && wpb == W && dp == P && fch == 1) return launch_bank_dma<N, FMA, T, W, P, 1>(center, job, st);
#if SG_DMA_EXPERIMENT == 2      // the short list (several half windows in one build)
    SG_DMA_TRY(32, 4, 12) SG_DMA_TRY(32, 8, 12) SG_DMA_TRY(32, 8, 16) SG_DMA_TRY(32, 4, 16) SG_DMA_TRY(32, 8, 8) SG_DMA_TRY(32, 16, 8)
#else
    SG_DMA_TRY(32, 4, 24) SG_DMA_TRY(32, 4, 32) SG_DMA_TRY(32, 2, 24) SG_DMA_TRY(32, 2, 32) SG_DMA_TRY(32, 8, 12) SG_DMA_TRY(32, 8, 16)
    SG_DMA_TRY(32, 4, 16) SG_DMA_TRY(32, 4, 8) SG_DMA_TRY(32, 4, 12) SG_DMA_TRY(32, 8, 8) SG_DMA_TRY(64, 4, 16) SG_DMA_TRY(64, 4, 8) SG_DMA_TRY(64, 4, 12)
    SG_DMA_TRY(48, 4, 8) SG_DMA_TRY(48, 4, 12) SG_DMA_TRY(96, 4, 12) SG_DMA_TRY(64, 8, 8) SG_DMA_TRY(64, 2, 12) SG_DMA_TRY(128, 4, 12)
    SG_DMA_TRY1(32, 4, 8) SG_DMA_TRY1(32, 4, 12) SG_DMA_TRY1(64, 4, 8) SG_DMA_TRY1(64, 4, 12) SG_DMA_TRY1(64, 4, 16) SG_DMA_TRY1(96, 4, 12) SG_DMA_TRY1(64, 8, 8)
    SG_DMA_TRY1(48, 4, 12) SG_DMA_TRY1(128, 4, 12) SG_DMA_TRY1(64, 2, 12)
#endif
#undef SG_DMA_TRY1
#undef SG_DMA_TRY
#endif
    // (waves per block, ring pairs) by half window and bank -- interleaved A/B on config 3's shape (profiles/r05_stream_shapes_ab.txt): light tiles run
    // best on fewer, deeper waves -- fused bank n = 8: (8, 12) 0.387 against (4, 12) 0.414 ms; bit-exact bank n = 8: (4, 16) 0.364 against 0.387.  The whole
    // library before / after this table, both banks: n = 4 / 6 / 8 2.5-6.5 % faster, fused n = 11 8.6 %, n = 2 and 10 level; the bit-exact bank from n = 11
    // (66 instructions per output pair at n = 16) keeps (4, 12): (8, 8) and (4, 16) measured 1-3 % slower there
#ifndef SG_DMA_FLAT_SHAPES        // (A/B builds: every tile on the default shape)
    if constexpr (SG_DMA_WPB == 4 && SG_DMA_PAIRS == 12 && SG_DMA_TR == 32) {
        if constexpr (N <= 5) return launch_bank_dma<N, FMA, 32, 4, 16>(center, job, st);
        else if constexpr (N <= 11 && FMA) return launch_bank_dma<N, FMA, 32, 8, 12>(center, job, st);
        else if constexpr (N <= 10) return launch_bank_dma<N, FMA, 32, 4, 16>(center, job, st);
    }
#endif
    return launch_bank_dma<N, FMA, SG_DMA_TR, SG_DMA_WPB, SG_DMA_PAIRS>(center, job, st);
}

template <int N>
static int dispatch_bank_dma(int n, int fma, const float *center, const BankJob &job, hipStream_t st)
{
    if (n == N) return fma ? launch_bank_dma_shape<N, true>(center, job, st) : launch_bank_dma_shape<N, false>(center, job, st);
    if constexpr (N < SG_DMA_MAX_N) return dispatch_bank_dma<N + 1>(n, fma, center, job, st);
    else return 1;
}

#ifndef SG_DMA_MIN_N
#define SG_DMA_MIN_N 1
#endif
#ifndef SG_DMA_FN
#define SG_DMA_FN sg_bank_dma_launch_all                     // the Makefile builds two objects (half windows 1..16 and 17..32) with a symbol each
#endif

// 0 = launched; 1 = not covered: half window outside [SG_DMA_MIN_N, SG_DMA_MAX_N], streams not a multiple of 128, rows not 16-byte aligned or
// 2 GiB and longer (the store descriptors), or fewer than two tiles of ticks
int SG_DMA_FN(int n, int fma, const float *center, const BankJob &job, int /*cu_count*/, hipStream_t st)
{
    if (n < SG_DMA_MIN_N || n > SG_DMA_MAX_N) return 1;
    if (job.streams % 128 != 0 || job.streams * 4 >= 0x7fffff00ull) return 1;
    if (((reinterpret_cast<uintptr_t>(job.samples) | reinterpret_cast<uintptr_t>(job.out) | reinterpret_cast<uintptr_t>(job.ring)) & 15u) != 0) return 1;
    if (job.ticks < 64) return 1;
    return dispatch_bank_dma<SG_DMA_MIN_N>(n, fma, center, job, st);
}

#else      // SG_DMA_MOM_BUILD: the third object, block-moment tiles of the fused bank
#ifndef SG_DMA_MOM_WPB
#define SG_DMA_MOM_WPB 8                                     // (waves per block, ring pairs) of the block-moment tiles: with a third of the arithmetic gone the tiles
#define SG_DMA_MOM_PAIRS 16                                  // want FEWER resident waves with deeper rings -- 8 waves per CU, 16 KiB in flight each (R5.9)
#endif
#ifndef SG_DMA_MOM_TR
#define SG_DMA_MOM_TR SG_DMA_TR                              // output ticks per block-moment tile
#endif

template <int N, int M, int TRT, int WPB, int DPR>
static int launch_bank_dma_mom(const StreamMomentFit &fit, const float *center, BankJob job, hipStream_t st)
{
    typedef DmaShape<N, TRT> D;
    typedef MomGeom<N> G;
    MomTaps<N, M> taps;
    memset(&taps, 0, sizeof(taps));
    auto put = [](f32x2 *v, int i, float x) { if (i & 1) v[i >> 1].y = x; else v[i >> 1].x = x; };
    for (int k = 0; k < 8; ++k) { put(taps.head, k, center[k]); put(taps.tail, k, center[2 * N - 7 + k]); }
    for (int sm = 0; sm < M; ++sm)
        for (int off = 0; off < G::NOFF; ++off) put(taps.c[sm], off, fit.c[sm][off]);
    for (int t = 0; t < 8; ++t) {
        const float q1 = (float)t - 3.5f;
        if (M > 1) put(taps.q[0], t, q1);
        if (M > 2) put(taps.q[M > 2 ? 1 : 0], t, q1 * q1 - 5.25f);
    }
    TileGeom geo;
    geo.strips = (unsigned)(job.streams / 128);
    geo.bands = (unsigned)((job.ticks + D::TR - 1) / D::TR);
    // strips per group: 32 KiB of a tick row (64 strips), a quarter of the row for narrow banks.  Measured over FRESH ALLOCATIONS inside one process
    // (tools/placement_stream.py, profiles/r05_placement_stream.txt): with 128-strip groups config 3's launch is 0.363-0.370 ms on most placements of the
    // two buffers and 0.405-0.414 on the rest (3 of 12 to 7 of 12 from call to call); with 64-strip groups 0.373-0.397 on all of them -- the same mean, a
    // third of the spread; 32 768 / 131 072 streams: 0.367 / 0.389 against 0.383 / 0.418; 16 384 streams: 32-strip groups 0.362 against 0.374.  Group sizes
    // that do not divide an XCD's eighth of the tile order (48, 80, 96) cost 10-18 %.  (The tap-by-tap tiles keep 128: the bit-exact bank is 5 % slower on 64.)
    unsigned want = geo.strips / 4;
    want = want < 16u ? 16u : (want > 64u ? 64u : want);
    geo.group = want;
    if (geo.group > geo.strips) geo.group = geo.strips;
    const unsigned groups = (geo.strips + geo.group - 1) / geo.group;
    geo.total = (unsigned long long)groups * geo.group * geo.bands;
    const unsigned long long blocks = (geo.total + WPB - 1) / WPB;
    if (geo.total >= 0x7fffff00ull) return 1;
    const unsigned grid = ((unsigned)blocks + 7u) & ~7u;
    job.aligned = 1;
    constexpr int DP = DPR < D::NI ? DPR : D::NI;
    constexpr size_t lds = (size_t)WPB * DP * 1024;
    static_assert(lds <= 160 * 1024, "slabs of one block must fit the CU's LDS");
    auto kernel = sg_bank_dma_kernel<N, true, TRT, WPB, DP, 1, M, MomTaps<N, M>>;
    if (lds > 65536 && hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {     // per launch: see launch_bank_dma
        (void)hipGetLastError();
        return 1;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * WPB), lds, st, job, taps, geo);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

template <int N, int M>
static int launch_bank_dma_mom_shape(const StreamMomentFit &fit, const float *center, const BankJob &job, hipStream_t st)
{
#ifdef SG_DMA_EXPERIMENT        // A/B build: waves per block and ring depth picked per process
    static const int tr = [] { const char *e = getenv("SAVGOL_HIP_STREAM_DMA_TR"); return e ? atoi(e) : SG_DMA_MOM_TR; }();
    static const int wpb = [] { const char *e = getenv("SAVGOL_HIP_STREAM_DMA_WPB"); return e ? atoi(e) : SG_DMA_MOM_WPB; }();
    static const int dp = [] { const char *e = getenv("SAVGOL_HIP_STREAM_DMA_PAIRS"); return e ? atoi(e) : SG_DMA_MOM_PAIRS; }();
#define SG_DMA_TRY(T, W, P) if (tr == T && wpb == W && dp == P) return launch_bank_dma_mom<N, M, T, W, P>(fit, center, job, st);
    SG_DMA_TRY(32, 8, 16) SG_DMA_TRY(32, 8, 12) SG_DMA_TRY(32, 4, 16)
    SG_DMA_TRY(48, 8, 16) SG_DMA_TRY(64, 8, 16) SG_DMA_TRY(64, 8, 12) SG_DMA_TRY(64, 8, 20) SG_DMA_TRY(64, 4, 16) SG_DMA_TRY(64, 4, 24) SG_DMA_TRY(64, 4, 32)
    SG_DMA_TRY(96, 8, 16) SG_DMA_TRY(128, 8, 16) SG_DMA_TRY(128, 4, 24)
#undef SG_DMA_TRY
#endif
    return launch_bank_dma_mom<N, M, SG_DMA_MOM_TR, SG_DMA_MOM_WPB, SG_DMA_MOM_PAIRS>(fit, center, job, st);
}

#ifndef SG_DMA_MOM_MIN_N
#define SG_DMA_MOM_MIN_N STREAM_MOMENT_MIN_N
#define SG_DMA_MOM_MAX_N STREAM_MOMENT_MAX_N
#endif
template <int N>
static int dispatch_bank_dma_mom(int n, const StreamMomentFit &fit, const float *center, const BankJob &job, hipStream_t st)
{
    if (n == N) {
        if (fit.terms == 1) return launch_bank_dma_mom_shape<N, 1>(fit, center, job, st);
        if (fit.terms == 2) return launch_bank_dma_mom_shape<N, 2>(fit, center, job, st);
        return launch_bank_dma_mom_shape<N, 3>(fit, center, job, st);
    }
    if constexpr (N < SG_DMA_MOM_MAX_N) return dispatch_bank_dma_mom<N + 1>(n, fit, center, job, st);
    else return 1;
}

// 0 = launched; 1 = not covered: half window outside 12..20, taps not a polynomial of degree <= 2 (stream_moment_fit), or a shape the tiles do not take
// (see below).  SAVGOL_HIP_STREAM_MOMENT=0: never (A/B runs against the tap-by-tap tiles).
int sg_bank_dma_launch_mom(int n, const float *center, const BankJob &job, int /*cu_count*/, hipStream_t st)
{
    static const int env = [] { const char *e = getenv("SAVGOL_HIP_STREAM_MOMENT"); return e ? atoi(e) : 1; }();
    if (!env || n < SG_DMA_MOM_MIN_N || n > SG_DMA_MOM_MAX_N) return 1;
    if (job.streams % 128 != 0 || job.streams * 4 >= 0x7fffff00ull) return 1;
    if (((reinterpret_cast<uintptr_t>(job.samples) | reinterpret_cast<uintptr_t>(job.out) | reinterpret_cast<uintptr_t>(job.ring)) & 15u) != 0) return 1;
    if (job.ticks < 64) return 1;
    StreamMomentFit fit;
    if (stream_moment_fit(n, center, &fit) == 0) return 1;
    // quadratic taps that sum to zero (a second derivative): the outputs are small against the samples, and what the fitted polynomial leaves of the
    // reference's fp32 taps (<= 3e-7 of the largest) shows in them -- 1.4e-6 of the oracle where the tap-by-tap tiles are at 0.7e-6 and the reference's
    // own loop at 0.6e-6 (tools/offset_probe_1d.py, R6.16).  Those banks keep the tap-by-tap tiles (7 % slower); config 3's taps are linear.
    if (job.centre && fit.terms >= 3) return 1;
    return dispatch_bank_dma_mom<SG_DMA_MOM_MIN_N>(n, fit, center, job, st);
}

#endif

}  // namespace sg
