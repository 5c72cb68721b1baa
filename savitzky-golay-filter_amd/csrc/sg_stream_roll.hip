// sg_stream_roll.hip -- savgol_streambank_push_block: rolling time windows in registers.
//
// A block push is a convolution down the time axis of a [tick][stream] array (history = ring contents, then this
// call's samples).  One WAVE owns 128 adjacent streams (a lane owns 2 = one 8-byte load per tick) and walks down a
// band of ticks.  No LDS, no __syncthreads, no re-reads except the 2n warm-up rows of a band.  Per output the
// arithmetic is the reference's (src/savgol_stream.c:166-185): one fp32 accumulator starting at 0, taps in ascending
// order, multiply and add rounded separately (v_pk_mul_f32 + v_pk_add_f32: two streams per instruction) ->
// bit-identical to savgol_stream_push for every stream.  Taps sit in SGPR pairs (by-value kernarg).
//   half windows <= 16: the last 2n+1 SAMPLES of the lane's streams live in a register ring that the fully unrolled
//                       tick loop indexes with literals; each tick is one 2n+1-term chain (bank_roll_item);
//   half windows 17-32: the unrolled ring would not fit the instruction cache, so the 2n in-flight ACCUMULATORS live
//                       in registers instead and every arriving sample advances all of them by one tap
//                       (bank_accroll_item; A/B at n <= 16: 15-30 % slower than the sample ring, so both stay).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <type_traits>
#include <utility>

#include "sg_internal.h"
#include "sg_pk.hpp"
#include "sg_runtime.hpp"
#include "sg_stream.hpp"
#include "sg_stream_roll.hpp"

namespace sg {

// One item: streams s0, s0+1 of every lane, output ticks t0 .. t0+nt-1.  Row r of the band = history index
// t0 - 2N + r; output tick m needs rows m .. m+2N, row r lives in ring slot r % U.
// Fused bank, derivative filters (job.centre; sg_stream_dma.hip has the tiles' form and the reasons, R6.16): what an item's rows are centred on -- the mean
// of eight real samples of each stream, spread from the oldest real one in the item's reach to its last row.  The bit-exact bank and smoothing filters: 0.
template <int N, bool VEC, bool FMA>
__device__ __forceinline__ f32x2 bank_item_centre(const BankJob &job, size_t s0, size_t t0, int nt, bool live0, bool live1)
{
    typedef SRoll<N> R;
    f32x2 cen = f32x2{0.0f, 0.0f};
    if constexpr (FMA) {
        if (job.centre) {                                    // uniform
            long long h0 = (long long)t0 - 2 * N;
            if (h0 < -(long long)job.received0) h0 = -(long long)job.received0;
            long long h1 = (long long)t0 + nt - 1;
            if (h1 > (long long)job.ticks - 1) h1 = (long long)job.ticks - 1;
            const long long span = h1 - h0;
            f32x2 sum = f32x2{0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long long h = h0 + (span * i) / 7;
                int slot = job.wp0 + (int)(h < 0 ? h : 0);
                slot = slot < 0 ? slot + R::WS : slot;
                const float *row = h >= 0 ? job.samples + (size_t)h * job.streams : job.ring + (size_t)slot * job.streams;
                if constexpr (VEC) sum = sum + *reinterpret_cast<const f32x2 *>(row + s0);
                else sum = sum + f32x2{live0 ? row[s0] : 0.0f, live1 ? row[s0 + 1] : 0.0f};
            }
            cen = sum * f32x2{0.125f, 0.125f};
            if (!(cen.x - cen.x == 0.0f)) cen.x = 0.0f;
            if (!(cen.y - cen.y == 0.0f)) cen.y = 0.0f;
        }
    }
    return cen;
}

template <int N, bool VEC, bool FMA>
__device__ __forceinline__ void bank_roll_item(const BankJob &job, const SRollTaps<N> &taps, size_t s0, size_t t0, int nt)
{
    typedef SRoll<N> R;
    const bool live0 = s0 < job.streams, live1 = s0 + 1 < job.streams;
    const f32x2 cen = bank_item_centre<N, VEC, FMA>(job, s0, t0, nt, live0, live1);      // fused bank, derivative filters: what the item's rows are centred on (else 0)
    const f32x2 backdt = cen * f32x2{job.centre_sum * job.dt_inv, job.centre_sum * job.dt_inv};
    auto load_row = [&](int r) -> f32x2 {
        long long h = (long long)t0 - 2 * N + r;
        if (h >= (long long)job.ticks) h = (long long)job.ticks - 1;            // past the call: loaded, never used
        int slot = job.wp0 + (int)(h < 0 ? h : 0);
        slot = slot < 0 ? slot + R::WS : slot;
        const float *row = h >= 0 ? job.samples + (size_t)h * job.streams : job.ring + (size_t)slot * job.streams;
        if constexpr (VEC) return *reinterpret_cast<const f32x2 *>(row + s0) - cen;
        else return f32x2{live0 ? row[s0] : 0.0f, live1 ? row[s0 + 1] : 0.0f} - cen;
    };
    f32x2 win[R::U];
#pragma unroll
    for (int r = 0; r < R::U - 1; ++r) win[r] = load_row(r);

    for (int base = 0; base < nt; base += R::U) {
        static_for(std::make_integer_sequence<int, R::U>{}, [&](auto uc) -> bool {
            constexpr int u = decltype(uc)::value;
            const int m = base + u;
            if (m >= nt) return false;                                           // uniform
            win[(u + R::U - 1) % R::U] = load_row(m + R::U - 1);                 // the slot row m-1 left
            // sum = 0; sum += w[k] * x[k], k ascending, separate roundings.  The product of tap k+1 is issued before
            // the add of tap k so that no instruction consumes the result of the one just before it.
            f32x2 acc = f32x2{0.0f, 0.0f};
            if constexpr (FMA) {
                // the fast form (SAVGOL_STREAMBANK_FMA): one v_pk_fma_f32 per tap instead of a multiply and an add -- half the
                // vector instructions.  Two chains (even taps, odd taps) so that no multiply-add waits for the one just before it;
                // not the reference's rounding (one rounding per term, two partial sums): within 1e-6 / 2e-6 of the fp64 oracle.
                f32x2 odd = pk_mul_sgpr<1>(taps.w[0], win[(u + 1) % R::U]);
                acc = pk_mul_sgpr<0>(taps.w[0], win[u % R::U]);
                static_for(std::make_integer_sequence<int, R::WS - 2>{}, [&](auto kc) -> bool {
                    constexpr int k = decltype(kc)::value + 2;
                    if constexpr (k & 1) pk_fma_sgpr<1>(odd, taps.w[k >> 1], win[(u + k) % R::U]);
                    else                 pk_fma_sgpr<0>(acc, taps.w[k >> 1], win[(u + k) % R::U]);
                    return true;
                });
                acc = acc + odd;
            } else {
            f32x2 p = pk_mul_sgpr<0>(taps.w[0], win[u % R::U]);
            static_for(std::make_integer_sequence<int, R::WS>{}, [&](auto kc) -> bool {
                constexpr int k = decltype(kc)::value;
                f32x2 pn = p;
                if constexpr (k + 1 < R::WS) pn = pk_mul_sgpr<((k + 1) & 1)>(taps.w[(k + 1) >> 1], win[(u + k + 1) % R::U]);
                acc = acc + p;
                p = pn;
                return true;
            });
            }
            const size_t t = t0 + (size_t)m;
            if (job.received0 + t + 1 >= (unsigned long long)R::WS) {            // uniform: an output exists (reference :166-170)
                const f32x2 y = FMA ? __builtin_elementwise_fma(acc, f32x2{job.dt_inv, job.dt_inv}, backdt) : acc * f32x2{job.dt_inv, job.dt_inv};
                float *orow = job.out + t * job.streams;
                if constexpr (VEC) __builtin_nontemporal_store(__builtin_bit_cast(u32x2, y), reinterpret_cast<u32x2 *>(orow + s0));   // written once (round 5: as the LDS-DMA tiles)
                else { if (live0) orow[s0] = y.x; if (live1) orow[s0 + 1] = y.y; }
            }
            return true;
        });
    }
}

// The same item, accumulator stationary (half windows above STREAM_RING_MAX_N, where the unrolled ring of the version
// above would not fit the instruction cache): 2N accumulator pairs instead of 2N+1 samples live in registers.  Slot a
// holds the output that has seen a samples so far; an arriving sample is tap a for slot a, and the add writes its sum
// into slot a+1 (walked from the top down, so that slot is already drained): every output still adds its taps in
// ascending order onto 0, with separate roundings -- and the tick loop is 2(2N+1) instructions, not unrolled.
template <int N, bool VEC, bool FMA>
__device__ __forceinline__ void bank_accroll_item(const BankJob &job, const SRollTaps<N> &taps, size_t s0, size_t t0, int nt)
{
    typedef SRoll<N> R;
    const bool live0 = s0 < job.streams, live1 = s0 + 1 < job.streams;
    const f32x2 cen = bank_item_centre<N, VEC, FMA>(job, s0, t0, nt, live0, live1);      // fused bank, derivative filters: what the item's rows are centred on (else 0)
    const f32x2 backdt = cen * f32x2{job.centre_sum * job.dt_inv, job.centre_sum * job.dt_inv};
    auto load_row = [&](int r) -> f32x2 {
        long long h = (long long)t0 - 2 * N + r;
        if (h >= (long long)job.ticks) h = (long long)job.ticks - 1;            // past the call: loaded, never used
        int slot = job.wp0 + (int)(h < 0 ? h : 0);
        slot = slot < 0 ? slot + R::WS : slot;
        const float *row = h >= 0 ? job.samples + (size_t)h * job.streams : job.ring + (size_t)slot * job.streams;
        if constexpr (VEC) return *reinterpret_cast<const f32x2 *>(row + s0) - cen;
        else return f32x2{live0 ? row[s0] : 0.0f, live1 ? row[s0 + 1] : 0.0f} - cen;
    };
    f32x2 acc[R::WS];                                        // acc[a], a = 1..2N; garbage until a real output reaches it
#pragma unroll
    for (int a = 0; a < R::WS; ++a) acc[a] = f32x2{0.0f, 0.0f};
    // rows in flight: a ring of PA registers that the row loop, unrolled PA times, indexes with literals.  (Until late in round 3
    // this was `ahead[p] = ahead[p + 1]` in a loop that was not unrolled: a register MOVE of a row still in flight has to wait for it,
    // so every iteration ended in s_waitcnt vmcnt(0) and nothing was ever ahead -- which is why these kernels lost to the sample
    // ring by 30-45 % and why more rows "ahead" made them slower.)
    // The loop is unrolled UA rows deep (hipcc drains every memory operation at the back edge: once per UA rows).
    // (ring depth x resident blocks swept in tools/r3/exp32.sh: four rows at full occupancy; eight from n = 28, +7 % at n = 32)
#ifdef SG_SROLL_PA
    constexpr int PA = SG_SROLL_PA, UA = N <= 24 ? 16 : 8;
#else
    constexpr int PA = N >= 28 ? 8 : 4, UA = N <= 24 ? 16 : 8;
#endif
    static_assert(UA % PA == 0, "slot = row % PA must carry over from one group of UA rows to the next");
    f32x2 ahead[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) ahead[p] = load_row(p);
    const int nrows = nt + 2 * N;
    for (int r0 = 0; r0 < nrows; r0 += UA) {
      static_for<UA>([&](auto uc) -> bool {
        constexpr int u = decltype(uc)::value;
        const int r = r0 + u;
        if (r >= nrows) return false;                        // uniform
        const f32x2 x = ahead[u % PA];
        ahead[u % PA] = load_row(r + PA);
        // volatile asm, products and adds alike: left to the compiler the adds sink to the end of the iteration and all
        // 2N+1 products stay live (see sg_2d_dense.hip)
        f32x2 done;
        static_for<R::WS>([&](auto ic) -> bool {
            constexpr int a = R::WS - 1 - decltype(ic)::value;   // slot = tap index, 2N down to 0
            if constexpr (FMA) {
                // fast form: slot a+1 = tap a * x + slot a in ONE instruction (the chain of an output stays a single chain)
                if constexpr (a == 0) {
                    acc[1] = pk_mul_sgpr<0>(taps.w[0], x);
                } else if constexpr (a == R::WS - 1) {
                    done = acc[a];
                    pk_fma_sgpr<(a & 1)>(done, taps.w[a >> 1], x);
                } else {
                    // acc[a + 1] = fma(w[a], x, acc[a]); three-address form so that the walk from the top down needs no copy
                    if constexpr ((a & 1) == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(acc[a + 1]) : "s"(taps.w[a >> 1]), "v"(x), "v"(acc[a]));
                    else                        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(acc[a + 1]) : "s"(taps.w[a >> 1]), "v"(x), "v"(acc[a]));
                }
                return true;
            }
            f32x2 p;
            if constexpr ((a & 1) == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "s"(taps.w[a >> 1]), "v"(x));
            else                        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(p) : "s"(taps.w[a >> 1]), "v"(x));
            if constexpr (a == R::WS - 1)  asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(done) : "v"(acc[a]), "v"(p));
            else if constexpr (a == 0)     asm volatile("v_pk_add_f32 %0, %1, 0 op_sel_hi:[1,0]" : "=v"(acc[1]) : "v"(p));
            else                           asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc[a + 1]) : "v"(acc[a]), "v"(p));
            return true;
        });
        const long long t = (long long)t0 + r - 2 * N;
        if (r >= 2 * N && job.received0 + (unsigned long long)t + 1 >= (unsigned long long)R::WS) {      // uniform (reference :166-170)
            const f32x2 y = FMA ? __builtin_elementwise_fma(done, f32x2{job.dt_inv, job.dt_inv}, backdt) : done * f32x2{job.dt_inv, job.dt_inv};
            float *orow = job.out + (size_t)t * job.streams;
            if constexpr (VEC) __builtin_nontemporal_store(__builtin_bit_cast(u32x2, y), reinterpret_cast<u32x2 *>(orow + s0));
            else { if (live0) orow[s0] = y.x; if (live1) orow[s0 + 1] = y.y; }
        }
        return true;
      });
    }
}

// ---- TILE form (round 4; half windows <= STREAM_RING_MAX_N, stream rows 16-byte aligned) ----
// The 2-D kernel's lesson (csrc/sg_2d_roll.hip, tools/membench_tile2d.hip): a walk down a strip -- a long-lived wave with a few rows in
// flight -- tops out at 0.63-0.71 of the roofline bare, short-lived waves that issue ALL their loads up front and exit reach 0.71-0.78.
// Here: a wave owns 256 adjacent streams (a lane 4 = one 16-byte load per tick row) and TR output ticks; it loads the TR + 2N rows
// (history from the ring, then this call's samples) back to back into registers, computes its TR rows with literal indices -- the same
// per-output arithmetic as bank_roll_item, bit for bit in both forms -- stores them and exits.  The 2N halo rows are read again by the
// tile of the next TR ticks: out of L2, because tiles are dealt so that blocks sharing an XCD walk down a GROUP of neighbouring strips
// band by band (a whole 256 KiB tick row of 65 536 streams would not stay in a 4 MiB L2 for the 256 tiles between two bands; a group of
// G strips x (TR + 2N) rows does).  Rows and stores go through range-checked buffer descriptors (streams beyond the bank read 0 and
// store nothing; a row without an output selects an empty descriptor), so there is no branch between the first load and the last store
// and hipcc's vmcnt waits stay counted.
#ifndef SG_STREAM_TILE_ROWS
#define SG_STREAM_TILE_ROWS 16
#endif
#ifndef SG_STREAM_TILE_WPB
#define SG_STREAM_TILE_WPB 2
#endif

#ifndef SG_STREAM_TILE_SPL
#define SG_STREAM_TILE_SPL 4                                 // streams per lane: 4 = one 16-byte load per row, 2 = one 8-byte load
#endif
template <int N, bool FMA>
__global__ __launch_bounds__(64 * SG_STREAM_TILE_WPB, 2) void sg_bank_tile_kernel(const BankJob job, const SRollTaps<N> taps, const TileGeom geo)
{
    typedef SRoll<N> R;
    constexpr int TR = SG_STREAM_TILE_ROWS, ROWS = TR + 2 * N, SPL = SG_STREAM_TILE_SPL, NP = SPL / 2;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nblk = gridDim.x;
    const unsigned blk = (job.aligned & 2) ? blockIdx.x : (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3);
    const unsigned long long t = (unsigned long long)blk * SG_STREAM_TILE_WPB + (unsigned)wv;
    if (t >= geo.total) return;
    // tile order: groups of `group` neighbouring strips; inside a group band after band, strips fastest
    const unsigned long long per_group = (unsigned long long)geo.group * geo.bands;
    const unsigned grp = (unsigned)(t / per_group);
    const unsigned long long rem = t % per_group;
    const unsigned gs = geo.strips - grp * geo.group < geo.group ? geo.strips - grp * geo.group : geo.group;     // strips in this (last) group
    const unsigned band = (unsigned)(rem / gs), strip = grp * geo.group + (unsigned)(rem % gs);
    if (band >= geo.bands) return;                                       // the last group is narrower: its tail of the t range is empty
    const size_t t0 = (size_t)band * TR;
    const unsigned voff = (strip * (64u * SPL) + (unsigned)SPL * (unsigned)lane) * 4u;     // byte offset of this lane's streams in a row
    const int row_bytes = (int)(job.streams * 4);

    struct Row { f32x2 p[NP]; };
    auto load_row = [&](int r) -> Row {
        long long h = (long long)t0 - 2 * N + r;
        if (h >= (long long)job.ticks) h = (long long)job.ticks - 1;            // past the call: loaded, never used
        int slot = job.wp0 + (int)(h < 0 ? h : 0);
        slot = slot < 0 ? slot + R::WS : slot;
        const float *row = h >= 0 ? job.samples + (size_t)h * job.streams : job.ring + (size_t)slot * job.streams;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(row), 0, row_bytes, 0x00020000);
        Row o;
        if constexpr (SPL == 4) {
            const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 0, 0));
            o.p[0] = f32x2{q.x, q.y}; o.p[1] = f32x2{q.z, q.w};
        } else {
            o.p[0] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, 0, 0));
        }
        return o;
    };
    Row win[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) win[r] = load_row(r);

    static_for<TR>([&](auto mc) -> bool {
        constexpr int m = decltype(mc)::value;
        f32x2 a[NP];
        if constexpr (FMA) {
            // two chains (even taps, odd taps), one v_pk_fma_f32 per tap and stream pair: bank_roll_item's fast form
            f32x2 o[NP];
#pragma unroll
            for (int c = 0; c < NP; ++c) { o[c] = pk_mul_sgpr<1>(taps.w[0], win[m + 1].p[c]); a[c] = pk_mul_sgpr<0>(taps.w[0], win[m].p[c]); }
            static_for(std::make_integer_sequence<int, R::WS - 2>{}, [&](auto kc) -> bool {
                constexpr int k = decltype(kc)::value + 2;
#pragma unroll
                for (int c = 0; c < NP; ++c) {
                    if constexpr (k & 1) pk_fma_sgpr<1>(o[c], taps.w[k >> 1], win[m + k].p[c]);
                    else                 pk_fma_sgpr<0>(a[c], taps.w[k >> 1], win[m + k].p[c]);
                }
                return true;
            });
#pragma unroll
            for (int c = 0; c < NP; ++c) a[c] = a[c] + o[c];
        } else {
            // the reference's order (src/savgol_stream.c:25-38): one accumulator from 0, taps ascending, multiply and add rounded separately
            f32x2 p[NP];
#pragma unroll
            for (int c = 0; c < NP; ++c) { a[c] = f32x2{0.0f, 0.0f}; p[c] = pk_mul_sgpr<0>(taps.w[0], win[m].p[c]); }
            static_for(std::make_integer_sequence<int, R::WS>{}, [&](auto kc) -> bool {
                constexpr int k = decltype(kc)::value;
#pragma unroll
                for (int c = 0; c < NP; ++c) {
                    f32x2 nx = p[c];
                    if constexpr (k + 1 < R::WS) nx = pk_mul_sgpr<((k + 1) & 1)>(taps.w[(k + 1) >> 1], win[m + k + 1].p[c]);
                    a[c] = a[c] + p[c];
                    p[c] = nx;
                }
                return true;
            });
        }
        const size_t tt = t0 + (size_t)m;
        const bool has_out = tt < job.ticks && job.received0 + tt + 1 >= (unsigned long long)R::WS;     // uniform (reference :166-170)
        const f32x2 s = f32x2{job.dt_inv, job.dt_inv};
        float *orow = job.out + (tt < job.ticks ? tt : 0) * job.streams;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(orow, 0, has_out ? row_bytes : 0, 0x00020000);
        if constexpr (SPL == 4) {
            const f32x2 y0 = a[0] * s, y1 = a[1] * s;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{y0.x, y0.y, y1.x, y1.y}), rs, (int)voff, 0, 0);
        } else {
            const f32x2 y0 = a[0] * s;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, y0), rs, (int)voff, 0, 0);
        }
        return true;
    });
}

template <int N, bool FMA>
__global__ __launch_bounds__(256) void sg_bank_roll_kernel(const BankJob job, const SRollTaps<N> taps)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // persistent waves; the 4 waves of a block take adjacent strips (2 KB contiguous per tick row)
    const unsigned nblk = gridDim.x;
    const unsigned blk = (job.aligned & 2) ? blockIdx.x : (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3);      // bit 1: launch order (A/B: SAVGOL_HIP_STREAM_XCD=0)
    const unsigned long long total = (unsigned long long)job.strips * job.bands;
    for (unsigned long long item = (unsigned long long)blk * 4u + (unsigned)wv; item < total; item += (unsigned long long)nblk * 4u) {
        const unsigned strip = (unsigned)(item % job.strips), band = (unsigned)(item / job.strips);
        const size_t s0 = (size_t)strip * 128 + 2 * (size_t)lane;
        const size_t t0 = (size_t)band * (size_t)job.band_ticks;
        const int nt = job.ticks - t0 < (size_t)job.band_ticks ? (int)(job.ticks - t0) : job.band_ticks;
        const bool vec = (job.aligned & 1) && (size_t)strip * 128 + 128 <= job.streams;
        if constexpr (N <= STREAM_RING_MAX_N) {
            if (vec) bank_roll_item<N, true, FMA>(job, taps, s0, t0, nt); else bank_roll_item<N, false, FMA>(job, taps, s0, t0, nt);
        } else {
            if (vec) bank_accroll_item<N, true, FMA>(job, taps, s0, t0, nt); else bank_accroll_item<N, false, FMA>(job, taps, s0, t0, nt);
        }
    }
}

template <int N, bool FMA>
static int launch_bank_roll(const float *center, BankJob job, int cu_count, hipStream_t st)
{
    typedef SRoll<N> R;
    SRollTaps<N> taps;
    memset(&taps, 0, sizeof(taps));
    for (int k = 0; k < R::WS; ++k) {
        if (k & 1) taps.w[k >> 1].y = center[k]; else taps.w[k >> 1].x = center[k];
    }
    static int per_cu = 0;                                   // resident blocks per CU of this instantiation
    if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sg_bank_roll_kernel<N, FMA>, 256, 0) != hipSuccess || nb < 1) nb = 2;
        per_cu = nb > 4 ? 4 : nb;
        if (FMA && N <= 16 && per_cu > 2) per_cu = 2;        // fewer waves, more rows in flight each (see SRoll::P)
    }
    // Where the tile form pays (profiles/r04_stream_tile.txt; config 3's shape, 65 536 streams x 4096 ... 16 384 ticks, five variants of
    // streams per lane x rows per tile x waves per block, strips per group 8 ... 256): the reference-order bank at n <= 12 (n = 4: 0.393 vs
    // 0.420 ms, n = 8: 0.408 vs 0.425); at n = 16 its 2n = 32 halo rows per 16-row tile (3 x the row reads out of L2) cancel the gain
    // (0.44-0.47 vs 0.45-0.46), and the fused multiply-add bank's walk is level or ahead at every half window (n = 16: 0.400-0.425 vs
    // 0.403-0.446).  Both forms sit at 0.60-0.69 of the roofline on this 2 GB call, a 0.4 ms launch, whatever the call's length.
    if constexpr (N <= 12 && !FMA) {
        // the tile form: rows of whole 16-byte quads (the buffer range check works on whole accesses), rows < 2 GiB, at least two tiles of ticks
        const bool quads = job.streams % 4 == 0 && job.streams * 4 < 0x7fffff00ull &&
                           ((reinterpret_cast<uintptr_t>(job.samples) | reinterpret_cast<uintptr_t>(job.out) | reinterpret_cast<uintptr_t>(job.ring)) & 15u) == 0;
        if (quads && job.ticks >= 2 * SG_STREAM_TILE_ROWS) {
            TileGeom geo;
            geo.strips = (unsigned)((job.streams + 64 * SG_STREAM_TILE_SPL - 1) / (64 * SG_STREAM_TILE_SPL));
            geo.bands = (unsigned)((job.ticks + SG_STREAM_TILE_ROWS - 1) / SG_STREAM_TILE_ROWS);
            // strips per group: (TR + 2N) rows x group KiB should stay well inside an XCD's 4 MiB L2 beside the rows in flight
            geo.group = (unsigned)(256 / SG_STREAM_TILE_SPL);      // 64 KiB of a tick row
            if (geo.group > geo.strips) geo.group = geo.strips;
            const unsigned groups = (geo.strips + geo.group - 1) / geo.group;
            geo.total = (unsigned long long)groups * geo.group * geo.bands;
            const unsigned long long blocks = (geo.total + SG_STREAM_TILE_WPB - 1) / SG_STREAM_TILE_WPB;
            if (blocks < 0x7fffff00ull) {
                unsigned grid = ((unsigned)blocks + 7u) & ~7u;
                job.aligned = 1;
                hipLaunchKernelGGL((sg_bank_tile_kernel<N, FMA>), dim3(grid), dim3(64 * SG_STREAM_TILE_WPB), 0, st, job, taps, geo);
                return 0;
            }
        }
    }
    const unsigned nwaves = (unsigned)cu_count * (unsigned)per_cu * 4u;
    job.strips = (unsigned)((job.streams + 127) / 128);
    // Bands of ticks: the resident waves take the strips x bands items round robin, so the call lasts (rounds of items) x (rows per band +
    // 2n warm-up rows).  Rounds 1-3 took ceil(waves / strips) bands, which is right when the strips divide the waves (65 536 streams: 512
    // strips, 4 bands, one round) and badly wrong next to it: 66 560 streams = 520 strips x 4 bands = 2080 items on 2048 waves -- a second
    // round for 32 items, 0.595 ms where 65 536 streams take 0.44.  Now the band count with the cheapest (rounds x rows) is taken
    // (a band re-reads 2n warm-up rows: keep it >= 8 windows).
    const size_t max_bands = job.ticks / (size_t)(8 * R::WS) > 0 ? job.ticks / (size_t)(8 * R::WS) : 1;
    size_t bands = 1;
    {
        // Candidates: every count up to the one that gives each resident wave an item (ceil(waves / strips), what rounds 1-3 took), bounded
        // by max_bands.  (round 4 stopped the search at 64, so a bank of FEW strips and a long block -- 1024 streams = 8 strips
        // x 100 000 ticks -- ran on 512 of its 2048 waves; the search is coarse above 64 to stay a few hundred steps.)
        double best = 1e300;
        const size_t fill = (nwaves + job.strips - 1) / (size_t)job.strips;
        size_t top = fill > 64 ? fill : 64;
        if (top > max_bands) top = max_bands;
        for (size_t b = 1; b <= top; b += (b < 64 ? 1 : (b / 64 < 1 ? 1 : b / 64))) {
            const size_t rows = (job.ticks + b - 1) / b + 2 * (size_t)N;
            const size_t rounds = ((size_t)job.strips * b + nwaves - 1) / nwaves;
            const double cost = (double)rounds * (double)rows;
            if (cost < best * 0.999) { best = cost; bands = b; }
        }
        if (top > 64) {                                      // the exact fill count is always a candidate
            const size_t b = top;
            const double cost = (double)(((size_t)job.strips * b + nwaves - 1) / nwaves) * (double)((job.ticks + b - 1) / b + 2 * (size_t)N);
            if (cost < best * 0.999) { best = cost; bands = b; }
        }
    }
    job.band_ticks = (int)((job.ticks + bands - 1) / bands);
    job.bands = (unsigned)((job.ticks + (size_t)job.band_ticks - 1) / (size_t)job.band_ticks);
    const unsigned long long total = (unsigned long long)job.strips * job.bands;
    unsigned grid = (unsigned)cu_count * (unsigned)per_cu;
    if ((unsigned long long)grid * 4ull > total) grid = (unsigned)((total + 3) / 4);
    grid = (grid + 7u) & ~7u;
    job.aligned = (job.streams % 2 == 0 && ((reinterpret_cast<uintptr_t>(job.samples) | reinterpret_cast<uintptr_t>(job.out) |
                                              reinterpret_cast<uintptr_t>(job.ring)) & 7u) == 0) ? 1 : 0;
    job.aligned |= 2;                                        // the walk's items in launch order (the XCD order lost, R3)
    hipLaunchKernelGGL((sg_bank_roll_kernel<N, FMA>), dim3(grid), dim3(256), 0, st, job, taps);
    return 0;
}

template <int N>
static int dispatch_bank_roll(int n, int fma, const float *center, const BankJob &job, int cu_count, hipStream_t st)
{
    if (n == N) return fma ? launch_bank_roll<N, true>(center, job, cu_count, st) : launch_bank_roll<N, false>(center, job, cu_count, st);
    if constexpr (N < STREAM_ROLL_MAX_N) return dispatch_bank_roll<N + 1>(n, fma, center, job, cu_count, st);
    else return 1;
}

// 0 = launched, 1 = half window not covered (the caller uses the LDS-tiled kernel).  ticks per call < 2^31 * band.
int sg_bank_roll_launch(int n, const float *center_weights, const float *ring, const float *samples, float *out, size_t streams,
                        int wp0, unsigned long long received0, size_t ticks, float dt_inv, int fma, int cu_count, hipStream_t st)
{
    if (n < 1 || n > STREAM_ROLL_MAX_N || ticks == 0 || ticks > (size_t)0x7fffffff) return 1;
    BankJob job;
    memset(&job, 0, sizeof(job));
    job.ring = ring; job.samples = samples; job.out = out;
    job.streams = streams; job.ticks = ticks; job.received0 = received0; job.wp0 = wp0; job.dt_inv = dt_inv;
    {
        double wsum = 0.0, wabs = 0.0;
        for (int k = 0; k <= 2 * n; ++k) { wsum += (double)center_weights[k]; wabs += std::fabs((double)center_weights[k]); }
        job.centre_sum = (float)wsum;
        job.centre = (fma && std::fabs(wsum) < 1e-3 * wabs) ? 1 : 0;       // smoothing filters (sum 1) gain nothing and a zero-mean stream would lose
    }
    // round 5: the LDS-DMA tile form where it covers the call (sg_stream_dma.hip; SAVGOL_HIP_STREAM_DMA=0 for A/B runs against the forms below)
    static const int dma_env = [] { const char *e = getenv("SAVGOL_HIP_STREAM_DMA"); return e ? atoi(e) : 1; }();
    if (dma_env && sg_bank_dma_launch(n, fma, center_weights, job, cu_count, st) == 0) return 0;
    return dispatch_bank_roll<1>(n, fma, center_weights, job, cu_count, st);
}

}  // namespace sg
