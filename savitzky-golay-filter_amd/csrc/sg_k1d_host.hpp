// sg_k1d_host.hpp -- what the host side needs to know about the 1-D kernels: the by-value kernel
// arguments and the launchers exported by the kernel objects.  No device code in here.
#pragma once

#include <hip/hip_runtime_api.h>

#include "sg_internal.h"

namespace sg {

// by-value kernarg -> s_load -> SGPRs.  66 floats = 33 aligned pairs: the fp32 kernel feeds whole
// SGPR pairs to v_pk_fma_f32 and picks the tap with op_sel.
struct alignas(8) Taps { union { float w[SAVGOL_MAX_WINDOW + 1]; double wd[(SAVGOL_MAX_WINDOW + 1) / 2]; }; };

// 16-byte vectors of output each lane owns (tile = 64 lanes x VPL vectors of one channel).  Host and kernels must
// agree.  The kernel supports 4, 6, 8, 12 and 16 (lane strides of VPL+1 vectors are conflict free for ds_read_b128);
// 8 -> 8 KiB tiles, 9.5 KB of LDS per wave, 4 waves per SIMD; 16 -> 16 KiB tiles, 2 waves per SIMD.
// Two tile widths are built where they differ, and the host picks per call (enqueue_batch):
//   narrow (vectors_per_lane): 8 everywhere.  4 / 6 / 8 are within 0.5 % of each other on big batches (tools/ab_1d.py).
//   wide (wide_vectors_per_lane): 16 for fp32 n <= 12 and fp64 n <= 24, 12 for fp32 n = 13..18 -- 2-5 % faster on big batches (32 GiB per launch,
//     A/B over eight placements of the buffers, tools/ab_1d_placements.py: fp32 n = 8: 5.76 -> 5.57 ms, fp64 n = 24: 5.81 -> 5.70;
//     n = 2: 5.81 -> 5.52, fp64 n = 20: 5.82 -> 5.49 in single-placement runs): fewer halo re-reads and fewer, longer waves.
//     Beyond those half windows the 64 accumulators and the window no longer fit (fp32 n = 23: 5 % slower, fp64 n = 28: 5 %
//     slower), and on a SMALL job wide tiles are simply fewer waves: one 10^6-sample signal (BASELINE config 1) takes 24.6 us on
//     245 wide tiles against 17.4 us on 489 narrow ones -- hence the choice by job size.
//     fp32 n = 13..18: 12 vectors (12 KiB tiles) instead: 2-6 % faster (n = 13: 5.75 -> 5.41 ms, n = 18: 5.72 -> 5.61 over eight
//     placements), 1-2 % slower at n = 20, 4-6 % at n = 21, 23; fp64 n >= 26: 12 is 3-4 % slower than 8, 16 level (+-1.5 %) with 8.
//   fp32 n >= 24 (block moments) is laid out for 32 outputs per lane: 8 either way.
#ifndef SG_VPL_F32_WIDE
#define SG_VPL_F32_WIDE 8          /* fp32, half_window >= 24; A/B builds override this */
#endif
#ifndef SG_VPL_NARROW
#define SG_VPL_NARROW 8            /* the narrow tile of every other kernel; A/B builds override this */
#endif
constexpr int vectors_per_lane(size_t elem_size, int half_window)
{
    return (elem_size == 4 && half_window >= 24) ? SG_VPL_F32_WIDE : SG_VPL_NARROW;
}
#ifndef SG_WIDE_F32_MAX16
#define SG_WIDE_F32_MAX16 12         /* last fp32 half window with 16-vector wide tiles (A/B builds override these three) */
#endif
#ifndef SG_WIDE_F32_MAX12
#define SG_WIDE_F32_MAX12 18         /* ... with 12-vector wide tiles above that */
#endif
#ifndef SG_WIDE_F64_MAX16
#define SG_WIDE_F64_MAX16 24         /* last fp64 half window with 16-vector wide tiles */
#endif
constexpr int wide_vectors_per_lane(size_t elem_size, int half_window)
{
#ifdef SG_VPL_NO_WIDE
    return vectors_per_lane(elem_size, half_window);                    /* A/B builds */
#else
    if (elem_size == 4 && half_window > SG_WIDE_F32_MAX16 && half_window <= SG_WIDE_F32_MAX12) return 12;
    return (elem_size == 8 ? half_window <= SG_WIDE_F64_MAX16 : half_window <= SG_WIDE_F32_MAX16) ? 16 : vectors_per_lane(elem_size, half_window);
#endif
}
// Tiles one launch may hold: the tile kernels run one tile per wave and four waves per 256-thread block, and HIP rejects a launch
// whose gridDim.x * blockDim.x reaches 2^32 (hip_runtime_api.h), i.e. 2^24 blocks.  Bigger jobs (tens of millions of short
// channels) are split over channels by the host.
constexpr unsigned MAX_TILES_PER_LAUNCH = 4u * ((1u << 24) - 8u);
// a job gets the wide tile when it has at least this many of them (8 rounds of the 2048 waves the chip holds at 2 per SIMD)
constexpr unsigned long long WIDE_TILE_MIN_TILES = 16384;

struct Job1D {
    const void *in;
    void       *out;
    long long   in_ld, out_ld;          // elements between channels
    unsigned    length;                 // samples per channel (< 2^31, as in the reference's int indexing)
    unsigned    tiles_per_channel;
    unsigned    total_tiles;            // channels * tiles_per_channel (< 2^31, the host splits bigger jobs)
    unsigned    tpc_magic, tpc_shift;   // tile / tiles_per_channel == umulhi(tile, tpc_magic) >> tpc_shift  (set_tiles_per_channel)
    unsigned    store_lo, store_hi;     // sample indices g whose result is stored ...
    unsigned    out_shift;              // ... at out[c*out_ld + g - out_shift]
    float       dt_inv;
    unsigned    flags;                  // FLAG_* below; boundary mode in the low byte
    // POLYNOMIAL edge rows ride along as extra items behind the tiles (sg1d_edge_item): item 2c = leading end of channel c,
    // 2c + 1 = trailing end; edges = the filter's [n][2n+1] edge table on the device, NULL = none
    unsigned    edge_items;
    const float *edges;
    // IN PLACE (out == in; round 6: two colour phases, no stash pass).  No tile may read what a neighbour has already overwritten.
    //   phase 1: the EVEN tiles of every channel.  Their neighbours' bodies are untouched, so their halos are read LIVE from the rows; only a halo that
    //            reaches past a channel end (tile 0's left, the last tile's right, and the right of the tile before a last tile shorter than NA) comes
    //            from `ends` ([channel][left of tile 0 | right of the last tile | right of tile T-2 | pad], NA samples each), where sg1d_launch_ends put
    //            it -- remapped per boundary mode -- before the phase.  Before storing, an even tile writes the first and last NA input samples of its
    //            own body into its odd neighbours' slots.
    //   phase 2: the ODD tiles: body live, both halos from their slot in `stash` ([channel][odd tile (k-1)/2][left NA | right NA]; an odd LAST tile's
    //            right half is written by sg1d_launch_ends).
    // tiles_per_channel / tpc_magic / total_tiles count the tiles of the PHASE; tpc_all = tiles of a whole channel.  The edge items read the 2n+1 samples
    // of their channel end from `edge_stash` ([channel][end][2n+1]).  NULL = out of place: halos and edge samples come from the rows themselves.
    const void *stash;
    const void *ends;
    const void *edge_stash;
    unsigned    phase, tpc_all;
    // tile order (sg1d_tile_body): 0 = each XCD sweeps one contiguous eighth of the tiles; s in 1..31 = chunks of 2^s blocks dealt to the XCDs round
    // robin (the eight fronts stay within 8 * 2^s blocks of each other); >= 32 = launch order.  SAVGOL_HIP_1D_XCD_CHUNK_LOG2, tools/placement_1d.py
    unsigned    xcd_chunk_log2;
    float       centre_sum;             // JOB_CENTRE: the sum of the reference's centre weights (what a constant input comes out as, before dt_inv)
};
// The fused strided (array-of-structs) kernel, sg1d_strided_kernel<N> (reference savgol_apply_strided, src/savgolFilter.c:877-934):
// sample i of channel c is the float at in + c * in_pitch + i * in_stride (bytes; the field offset is folded into `in`), all
// 4-byte aligned.  Same tiles, same slab and the same inner product as the dense kernel; only the staging loads and the
// stores walk records.
struct JobStrided {
    const char *in;
    char       *out;
    long long   in_pitch, out_pitch;    // bytes between channels
    long long   in_stride, out_stride;  // bytes between elements
    unsigned    length;
    unsigned    tiles_per_channel, total_tiles, tpc_magic, tpc_shift;
    unsigned    store_lo, store_hi;     // sample indices whose result is stored
    float       dt_inv;
    unsigned    flags;                  // boundary mode in the low byte, JOB_SCALE, JOB_EDGE_NEGATE
    unsigned    edge_items;             // as in Job1D
    const float *edges;
};
// Division by an invariant on the scalar unit (gfx950 has s_mul_hi_u32 but no scalar divide; left as `/` the compiler
// runs the float-reciprocal sequence on the VECTOR unit, ~25 instructions per tile in a kernel that is VALU-issue bound).
// For 0 <= t < 2^31 and 2^(l-1) < d <= 2^l:  floor(t / d) == (t * ceil(2^(31+l) / d)) >> (31 + l), and the multiplier
// fits 32 bits; d == 1 is flagged with shift 32.
inline void division_magic(unsigned d, unsigned *magic, unsigned *shift)
{
    if (d <= 1) { *magic = 0; *shift = 32; return; }
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    const unsigned long long p = 1ull << (31 + l);
    *magic = (unsigned)((p + d - 1) / d);
    *shift = l - 1;
}
inline void set_tiles_per_channel(Job1D &job, unsigned d)
{
    job.tiles_per_channel = d;
    division_magic(d, &job.tpc_magic, &job.tpc_shift);
}

// Block-moment kernels, common constants.  One device table per filter content, read through scalar loads; OFF = the slab offset of the first sample
// output 0 reads (K1D<float, n>).  (Round 2's whole-lane fp32 form, sg_k1d_moment.hpp, went in round 6; its plan slot size stays.)
constexpr int MOMENT_MIN_N = 24, MOMENT_MAX_N = 32;         // the fp64 kernel's range; the fp32 half-lane form starts at MOMENTH_MIN_N
constexpr int MOMENT_MAX_TERMS = 7;
constexpr int MOMENT_TABLE_FLOATS = 400;
constexpr int moment_off(int n) { return (n + 3) / 4 * 4 - n; }
struct MomentArgs { const float *table; };

// The fp32 form on HALF-lane blocks (sg_k1d_momenth.hpp, round 5): a lane's 32 outputs as two groups of 16, each with the block X[LO .. HI) of ITS window
// (LO even >= 15 + OFF, HI even <= OFF + 2n + 1: 48 samples at n = 32), paired front to back, two pairs per packed step.
//   floats [0, 132)    tap pairs (w[k], w[k-1]), k = 0 .. 2n+1, w[-1] = w[2n+1] = 0   (what one broadcast sample feeds into an output pair)
//   floats [132, 276)  phi pairs [u][s-1] = (phi_s(2u), phi_s(2u+1)), u = 0 .. BK/4-1 (<= 12 steps), s = 1..6: six pairs per step
//   floats [276, 388)  c pairs [s][j] = (c_s(2j), c_s(2j+1)), s = 0..6, j = 0..7
constexpr int MOMENTH_MIN_N = 20;                           // A/B in one process (profiles/r05_ab_momenth.txt): n = 18 the plain sum wins by 3 %, 20 / 22 / 23 this form by 0.6 / 1.1 / 1.4 %, 32 by 2.3-4 %
constexpr int MOMENTH_MAX_STEPS = 12;
constexpr int MOMENTH_OFF_W = 0, MOMENTH_OFF_PHI = 132, MOMENTH_OFF_C = MOMENTH_OFF_PHI + MOMENTH_MAX_STEPS * 12;
constexpr int MOMENTH_TABLE_FLOATS = MOMENTH_OFF_C + MOMENT_MAX_TERMS * 16;
static_assert(MOMENTH_TABLE_FLOATS <= MOMENT_TABLE_FLOATS, "the half-lane table travels in the same plan slot");
constexpr int momenth_lo(int n) { return (15 + moment_off(n) + 1) / 2 * 2; }
constexpr int momenth_hi(int n) { return (moment_off(n) + 2 * n + 1) / 2 * 2; }

// The fp64 block-moment path (sg_k1d_moment64.hpp, half windows 24..32; savgol_apply_batch_f64_tol with rel_tol >= 1e-6, or SAVGOL_BATCH_MOMENT_F64): 16 outputs per lane, the window is
// X[0 .. 16 + 2n + OFF), the common block X[LO .. HI) with LO = 15 + OFF, HI = OFF + 2n + 1 (2n - 14 samples: 50 at n = 32), paired front to back.
//   doubles [0, 16)     centre taps 0..14 (exact promotions of the fp32 table), one pad
//   doubles [16, 166)   phi[t][s-1] = P_s((t - (BK-1)/2) / (BK/2)), t = 0..BK/2-1 (<= 25 pairs), s = 1..6: six per pair whatever the term count
//   doubles [166, 278)  c[s][r], s = 0..6, r = 0..15: the block's share of output r is sum_s c_s(r) mu_s
constexpr int MOMENT64_MAX_PAIRS = 25;
constexpr int MOMENT64_OFF_W = 0, MOMENT64_OFF_PHI = 16, MOMENT64_OFF_C = MOMENT64_OFF_PHI + MOMENT64_MAX_PAIRS * 6;
constexpr int MOMENT64_TABLE_DOUBLES = MOMENT64_OFF_C + MOMENT_MAX_TERMS * 16;
constexpr int moment64_off(int n) { return (n + 1) / 2 * 2 - n; }                     // OFF of K1D<double, n>
constexpr int moment64_lo(int n) { return 15 + moment64_off(n); }
constexpr int moment64_hi(int n) { return moment64_off(n) + 2 * n + 1; }
struct Moment64Args { const double *table; };

enum : unsigned {
    JOB_MODE_MASK  = 0xffu,             // SavgolBoundaryMode value; 0/unknown: out-of-range reads are 0
    JOB_SCALE      = 1u << 8,           // multiply by dt_inv (dt_inv != 1)
    JOB_VEC_IN     = 1u << 9,           // input rows are 16-B aligned
    JOB_VEC_OUT    = 1u << 10,          // output rows (after out_shift) are 16-B aligned
    JOB_ODD_TAPS   = 1u << 11,          // fp64: taps.wd holds taps 0..n, tap 2n-k = -tap k (odd derivative) instead of +tap k
    JOB_EDGE_NEGATE = 1u << 12,         // edge items: negate the leading-edge outputs (SAVGOL_BATCH_CORRECT_LEADING_EDGE, odd derivatives)
    JOB_CENTRE     = 1u << 13,          // fp32 derivative filters: the tile's samples are centred on its first body sample before the inner products,
                                        //   c * centre_sum is added back (sg1d_tile_body; R6.16)
};

}  // namespace sg

extern "C" {
// one object per (type, half-window group), see the Makefile; each returns 1 if it owns n
int sg1d_launch_f32_g0(int n, int wide, const sg::Job1D *job, const sg::Taps *taps, unsigned grid, void *stream);
int sg1d_launch_f32_g1(int n, int wide, const sg::Job1D *job, const sg::Taps *taps, unsigned grid, void *stream);
int sg1d_launch_f32_g2(int n, int wide, const sg::Job1D *job, const sg::Taps *taps, unsigned grid, void *stream);
int sg1d_launch_f32_g3(int n, int wide, const sg::Job1D *job, const sg::Taps *taps, unsigned grid, void *stream);
// fused strided kernel, fp32, one launcher per half-window group (same objects as the dense fp32 kernels); 1 if this group owns n
int sg1d_launch_strided_f32_g0(int n, const sg::JobStrided *job, const sg::Taps *taps, unsigned grid, void *stream);
int sg1d_launch_strided_f32_g1(int n, const sg::JobStrided *job, const sg::Taps *taps, unsigned grid, void *stream);
int sg1d_launch_strided_f32_g2(int n, const sg::JobStrided *job, const sg::Taps *taps, unsigned grid, void *stream);
int sg1d_launch_strided_f32_g3(int n, const sg::JobStrided *job, const sg::Taps *taps, unsigned grid, void *stream);
int sg1d_launch_f64_g0(int n, int wide, const sg::Job1D *job, const sg::Taps *taps, unsigned grid, void *stream);
int sg1d_launch_f64_g1(int n, int wide, const sg::Job1D *job, const sg::Taps *taps, unsigned grid, void *stream);
int sg1d_launch_f64_g2(int n, int wide, const sg::Job1D *job, const sg::Taps *taps, unsigned grid, void *stream);
int sg1d_launch_f64_g3(int n, int wide, const sg::Job1D *job, const sg::Taps *taps, unsigned grid, void *stream);

// the fp32 kernel for half window n (24..32) with `terms` block moments (3, 5 or 7); one object per term count; returns 0 when enqueued
// fits the polynomial behind the 2n+1 centre taps and fills table[MOMENT_TABLE_FLOATS]; returns the number of moments the kernel
// needs (3, 5, 7) or 0 when n is outside 24..32 or the taps are not a polynomial of degree <= 6 to fp32 rounding
// (sg_k1d_moment_fit.cpp)
// the half-lane form's table (MOMENTH_TABLE_FLOATS floats) and launchers; same return value
int sg1d_momenth_prepare(int n, const float *center_weights, float *table);
int sg1d_launch_f32_momenth_t3(int n, const sg::Job1D *job, const float *d_table, unsigned grid, void *stream);
int sg1d_launch_f32_momenth_t5(int n, const sg::Job1D *job, const float *d_table, unsigned grid, void *stream);
int sg1d_launch_f32_momenth_t7(int n, const sg::Job1D *job, const float *d_table, unsigned grid, void *stream);
// the opt-in fp64 counterpart: table[MOMENT64_TABLE_DOUBLES]; same return value
int sg1d_moment64_prepare(int n, const float *center_weights, double *table);
int sg1d_launch_f64_moment_t3(int n, const sg::Job1D *job, const double *d_table, unsigned grid, void *stream);
int sg1d_launch_f64_moment_t5(int n, const sg::Job1D *job, const double *d_table, unsigned grid, void *stream);
int sg1d_launch_f64_moment_t7(int n, const sg::Job1D *job, const double *d_table, unsigned grid, void *stream);

// in place, before the two phases: the halos that reach past a channel end (Job1D::ends; an odd last tile's right half goes into its stash slot too),
// remapped per boundary mode, and the edge rows' samples -- a few hundred samples per channel
int sg1d_launch_ends(const void *in, long long in_ld, unsigned length, unsigned tiles_per_channel, int TW, int NA, int mode, void *stash, void *ends, void *edge_stash,
                     int ws, size_t channels, int elem_bytes, void *stream);
int sg1d_launch_reference_order_f32(const float *in, float *out, long long in_ld, long long out_ld, long long L, int n,
                                    const float *d_table, float dt_inv, int mode, int store_lo, int store_hi, int out_shift,
                                    int negate_leading, size_t channels, void *stream);
int sg1d_launch_refpk_f32(const float *in, float *out, long long in_ld, long long out_ld, long long length, int n,
                          const float *center, float dt_inv, int mode, int store_lo, int store_hi, int out_shift,
                          size_t channels, int cu_count, void *stream);
int sg_launch_gather_f32(const void *base, size_t stride, size_t offset, size_t pitch, float *dst, size_t dst_ld,
                         size_t channels, size_t count, void *st);
int sg_launch_scatter_f32(const float *src, size_t src_ld, void *base, size_t stride, size_t offset, size_t pitch,
                          size_t channels, size_t count, void *st);
}

namespace sg {

template <typename T> int launch_center(int n, int wide, const Job1D &job, const Taps &taps, unsigned grid, hipStream_t st);
template <> inline int launch_center<float>(int n, int wide, const Job1D &job, const Taps &taps, unsigned grid, hipStream_t st)
{
    const int hit = sg1d_launch_f32_g0(n, wide, &job, &taps, grid, st) || sg1d_launch_f32_g1(n, wide, &job, &taps, grid, st) ||
                    sg1d_launch_f32_g2(n, wide, &job, &taps, grid, st) || sg1d_launch_f32_g3(n, wide, &job, &taps, grid, st);
    if (!hit) { sg_set_error("no fp32 kernel for half_window %d", n); return -1; }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sg_set_error("1-D kernel launch failed: %s", hipGetErrorString(e)); return -1; }
    return 0;
}
template <> inline int launch_center<double>(int n, int wide, const Job1D &job, const Taps &taps, unsigned grid, hipStream_t st)
{
    const int hit = sg1d_launch_f64_g0(n, wide, &job, &taps, grid, st) || sg1d_launch_f64_g1(n, wide, &job, &taps, grid, st) ||
                    sg1d_launch_f64_g2(n, wide, &job, &taps, grid, st) || sg1d_launch_f64_g3(n, wide, &job, &taps, grid, st);
    if (!hit) { sg_set_error("no fp64 kernel for half_window %d", n); return -1; }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sg_set_error("1-D kernel launch failed: %s", hipGetErrorString(e)); return -1; }
    return 0;
}

inline int launch_strided(int n, const JobStrided &job, const Taps &taps, unsigned grid, hipStream_t st)
{
    const int hit = sg1d_launch_strided_f32_g0(n, &job, &taps, grid, st) || sg1d_launch_strided_f32_g1(n, &job, &taps, grid, st) ||
                    sg1d_launch_strided_f32_g2(n, &job, &taps, grid, st) || sg1d_launch_strided_f32_g3(n, &job, &taps, grid, st);
    if (!hit) { sg_set_error("no strided kernel for half_window %d", n); return -1; }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sg_set_error("strided kernel launch failed: %s", hipGetErrorString(e)); return -1; }
    return 0;
}


}  // namespace sg
