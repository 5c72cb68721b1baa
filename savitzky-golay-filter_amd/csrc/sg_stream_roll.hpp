// sg_stream_roll.hpp -- types shared by the block-push kernels (sg_stream_roll.hip: walk and register tiles; sg_stream_dma.hip: LDS-DMA tiles)
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

#include "sg_pk.hpp"
#include "sg_stream.hpp"
#include "sg_stream_host.hpp"

namespace sg {

template <int N>
struct SRoll {
    static constexpr int WS = 2 * N + 1;
    // rows loaded ahead of the arithmetic (A/B builds override).  Round 3, after the counters had said that a walk is short of
    // requests in flight rather than of memory (profiles/r03_strip_walk_counters.txt): the sample-ring kernels (n <= 16) with 7 rows
    // ahead -- and, for the fused-multiply-add bank, at most TWO resident blocks per CU (8 waves; launch_bank_roll) -- run config 3's
    // block push in 0.389 ms instead of 0.404-0.417 (0.69 of the roofline; n = 4: 0.378 vs 0.407), the reference-order bank 0.451-0.460
    // instead of 0.467-0.469 at its full occupancy (it is bound by its two instructions per tap and needs the waves).  The
    // accumulator-ring kernels (n > 16) have their own ring of rows in flight (bank_accroll_item) and keep 4 blocks per CU.
#ifdef SG_SROLL_P
    static constexpr int P = SG_SROLL_P;
#else
    static constexpr int P = N <= 16 ? 7 : 3;
#endif
    static constexpr int U = WS + P;                         // ring slots = unroll factor of the tick loop
    static constexpr int NP = N + 1;                         // SGPR pairs holding taps 0..2N
};

template <int N>
struct SRollTaps { f32x2 w[SRoll<N>::NP]; };

struct BankJob {
    const float *ring;               // [WS][streams], slot (wp0 - k) mod WS = sample -k of the history
    const float *samples;            // [ticks][streams]
    float       *out;                // [ticks][streams]
    size_t       streams, ticks;
    unsigned long long received0;    // samples per stream before this call
    int          wp0;
    float        dt_inv;
    unsigned     strips, bands;
    int          band_ticks;
    int          aligned;            // rows of samples / ring / out start 8-byte aligned (streams even, bases aligned)
    float        centre_sum;         // fused bank, LDS-DMA tiles: the sum of the reference's centre weights ...
    int          centre;             // ... and 1 when that sum is (nearly) zero -- a derivative filter: the tiles then run on centred samples (sg_stream_dma.hip)
};

struct TileGeom { unsigned strips, bands, group; unsigned long long total; };

// sg_stream_dma.hip: the LDS-DMA tile form of the block push (round 5).  0 = launched, 1 = not covered (the caller walks)
int sg_bank_dma_launch_lo(int n, int fma, const float *center, const BankJob &job, int cu_count, hipStream_t st);      // half windows 1..16
int sg_bank_dma_launch_hi(int n, int fma, const float *center, const BankJob &job, int cu_count, hipStream_t st);      // 17..32
// Which half windows take the LDS-DMA tiles (profiles/r05_stream_dma.txt, config 3's shape, sustained): every n <= 16; above 16 the FMA bank gains
// 6 % (n = 17) ... 15 % (n = 32) over the accumulator-ring walk and the bit-exact bank 10-19 % from n = 24, while at n = 17 its sustained time is
// 10 % worse (0.603 against 0.548 ms: twice the vector instructions, and the chip lowers its clock under them) -- it keeps the walk below 20.
// the fused bank where its taps are a polynomial of degree <= 2 (config 3: linear): blocks of 8 ticks through their moments; 1 = not covered
int sg_bank_dma_launch_mom(int n, const float *center, const BankJob &job, int cu_count, hipStream_t st);              // half windows 12..20
inline int sg_bank_dma_launch(int n, int fma, const float *center, const BankJob &job, int cu_count, hipStream_t st)
{
    if (fma && n >= STREAM_MOMENT_MIN_N && n <= STREAM_MOMENT_MAX_N && sg_bank_dma_launch_mom(n, center, job, cu_count, st) == 0) return 0;
    if (n <= 16) return sg_bank_dma_launch_lo(n, fma, center, job, cu_count, st);
    if (!fma && n < 20) return 1;
    return sg_bank_dma_launch_hi(n, fma, center, job, cu_count, st);
}

}  // namespace sg
