/*
 * savgol_hip.h -- device-side C ABI of the MI355X Savitzky-Golay library (libsavgol_hip.so).
 *
 * The reference API (savgolFilter.h, savgol_stream.h, savgol2d.h) is one signal / one stream /
 * one image at a time, fp32, host pointers.  The workloads this library is built for -- thousands
 * of channels resident in HBM, tens of thousands of concurrent streams, image stacks, fp64 --
 * cannot be expressed through it, so this header ADDS entry points next to the drop-in ones.
 * Everything here is plain C: raw device pointers, sizes, an opaque `void *stream`
 * (a hipStream_t; NULL = the default stream).  No torch / C++ types cross this boundary.
 *
 * Each function names the reference routine whose arithmetic it performs.  All of them return
 * 0 on success and -1 on error unless stated otherwise; savgol_hip_last_error() has the text.
 * Launches are asynchronous on `stream`.  What "enqueue only" means, entry point by entry point:
 *   - savgol_apply[_valid]_batch_f32/f64, savgol2d_apply_batch_f32, savgol2d_gradient/hessian/laplacian_batch_f32 (square and
 *     rectangular windows), savgol_streambank_push/_push_full/_push_block/_flush/_flush_leading/_reset: launches only --
 *     capturable into a hipGraph AFTER one warm-up call with the same filter: the FIRST call with a new filter content
 *     uploads its tables (hipMalloc + a synchronous hipMemcpy, then cached for the life of the process; tables are never
 *     freed or moved, so captured graphs and queued launches stay valid) and, for the derivative frames, solves the
 *     least-squares problem on the host (cached per configuration);
 *   - savgol_apply_strided_batch_f32[_ex]: launches only when the fields are 4-byte aligned and disjoint (the fused kernel);
 *     otherwise (reference summation order, unaligned or overlapping fields) launches plus a stream-ordered allocation /
 *     free pair for its two dense frames, from the library's own retained memory pool (no shared scratch, no lock, no
 *     synchronise); savgol2d_apply_rowband_f32 and channels longer than 2^30 samples use the same pool for their scratch;
 *   - savgol_streambank_save/_load, savgol_hip_synchronize and every host-pointer drop-in call of savgolFilter.h /
 *     savgol_stream.h / savgol2d.h: synchronous by nature (they return host data).
 * Short host-pointer calls (savgol_apply / _valid / _strided on <= 4096 samples and <= 64 K multiply-adds, every savgol_stream_* call
 * on one SavgolStream) do not launch: one workgroup stays resident behind a doorbell (csrc/sg_k1d_misc.hip) and answers in 6-8 us
 * with the reference's bits.  It leaves by itself after 60 us without a call -- a device-wide synchronise issued by the caller
 * waits at most that long for it -- is told to leave before the library's own hipFree / hipMalloc, is restarted by the next short
 * call, and after an unanswered call stays away for 1 s, doubling to 64 s.  Environment: SAVGOL_HIP_SMALL_SERVICE=0 disables it
 * (every call then launches), SAVGOL_HIP_SMALL_SERVICE_IDLE_US sets the idle time (50 ... 5 000 000).  The library installs no
 * signal handlers unless SAVGOL_HIP_BAR_DOORBELL=1 asks for doorbells in BAR-mapped device memory (probed with a guarded store).
 * Accuracy of the default fp32 device kernels against the double-accumulation oracle (normwise, max|err| / max|ref|): ONE rule, everywhere --
 * <= max(1e-6, 1.1 x the error of the reference's OWN fp32 arithmetic on the same samples), i.e. 1e-6 outright wherever the reference itself
 * meets 1e-6 (smoothing filters that pass the signal: always).  1-D: three interleaved partial sums per output, block moments from half window
 * 20; 2-D: the pass whose taps cancel harder runs first (derivative frames included: no wider constant since round 6).  tests/_util.py holds the
 * rule, tools/parity_margins.py prints every comparison's margin.  Bit-identical-to-the-reference results:
 * SAVGOL_HIP_OPT_REFERENCE_SUMMATION (1-D), method 1 (2-D), and every host-pointer drop-in call.
 */
#ifndef SAVGOL_HIP_H
#define SAVGOL_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include "savgolFilter.h"
#include "savgol2d.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- runtime ------------- */
int         savgol_hip_device_count(void);          /* usable HIP devices (0 = none)       */
int         savgol_hip_set_device(int ordinal);     /* device used by THIS thread's calls  */
int         savgol_hip_get_device(void);
int         savgol_hip_synchronize(void *stream);       /* hipStreamSynchronize; the scratch pool keeps its threshold (savgol_hip_trim_scratch hands it back) */
/* Calls that need a temporary frame (staged strided path, reference-order batches, channel ends of very long channels, row-band strips)
 * take it from the library's own stream-ordered pool, which keeps up to 256 MiB of freed blocks for the next call
 * (SAVGOL_HIP_SCRATCH_KEEP_MB).  This returns every unused byte of it to the driver now.  0 / -1. */
int         savgol_hip_trim_scratch(void);
size_t      savgol_hip_scratch_reserved(void);          /* bytes the pool holds right now (hipMemPoolAttrReservedMemCurrent) */
/* Multi-GPU: the path shards by independent units (channels, streams, images) with no data-path collective, so the whole
 * "context" is: one process (or thread) per GPU calls savgol_hip_set_device(local_rank) and filters its own contiguous slice.
 * This returns that slice [*lo, *hi) of `total` units for `rank` of `world_size` (the first total % world_size ranks take
 * one unit more); -1 on bad arguments.  bench.py and the Python mirror use the same arithmetic.                        */
int         savgol_hip_shard_range(size_t total, int world_size, int rank, size_t *lo, size_t *hi);
const char *savgol_hip_last_error(void);            /* thread-local, never NULL            */
const char *savgol_hip_version(void);
/* Process-wide switches; defaults reproduce the reference bit for bit in behaviour, quirks included.
 * SAVGOL_HIP_OPT_CORRECT_LEADING_EDGE = 1: in POLYNOMIAL mode the first n outputs of an ODD derivative get the
 * mathematically correct sign.  (The reference applies the trailing-edge rows to reversed data, which negates odd
 * derivatives on the leading edge -- src/savgolFilter.c:773-777)  Affects the 1-D batch / apply
 * entry points only; the streaming path keeps the reference behaviour.                                           */
enum { SAVGOL_HIP_OPT_CORRECT_LEADING_EDGE = 1, SAVGOL_HIP_OPT_REFERENCE_SUMMATION = 2, SAVGOL_HIP_OPT_PLAIN_SUMMATION = 3,
       SAVGOL_HIP_OPT_BOUNDARY_AWARE = 4, SAVGOL_HIP_OPT_TILE_WIDTH = 5 };
/* SAVGOL_HIP_OPT_REFERENCE_SUMMATION = 1: the fp32 1-D batch / valid / strided DEVICE entry points sum each output in
 * the reference's own order (convolve_ilp, src/savgolFilter.c:547-580: four chains, separate multiply and add) and are
 * then bit-identical to the reference's savgol_apply; 1.5x (n=5) to 2.1x (n=32) slower than the default FMA kernel, which
 * agrees with it to 1e-6.  The host-pointer drop-in calls of savgolFilter.h always use that order.
 * SAVGOL_HIP_OPT_PLAIN_SUMMATION = 1: the fp32 batch kernels apply all 2n+1 taps one by one at every half window.  By default, from half
 * window 20 up (poly_order >= 2, derivative <= 1), a lane treats its 32 outputs as two groups of 16 and replaces the taps that fall on the
 * samples common to a group's windows by block moments (the taps are a polynomial of degree <= poly_order in the tap index;
 * csrc/sg_k1d_momenth.hpp): 19 instead of 33 packed multiply-adds per output pair at half_window 32, the same 1e-6 agreement with the fp64
 * oracle, different last bits.  Filters whose table is not such a polynomial (hand-edited center_weights), poly_order > 6, poly_order < 2,
 * second derivatives and half windows below 20 always use the plain sum.
 * SAVGOL_HIP_OPT_BOUNDARY_AWARE = 1: the paths that IGNORE config.boundary in the reference honour it:
 *   - savgol_apply_strided / savgol_apply_strided_batch_f32 (reference src/savgolFilter.c:877-934 always uses the polynomial
 *     edge rows) apply the configured mode, like savgol_apply;
 *   - savgol_stream_* and savgol_streambank_* (reference src/savgol_stream.c:43-74 likewise) emit REFLECT / CONSTANT edges in
 *     push_full / flush / flush_leading: the first and last n outputs are the centre taps on the index-remapped window
 *     (get_padded_sample, :442-482), so push_full... + flush equals savgol_apply in that mode.  PERIODIC needs samples from the
 *     other end of the signal and keeps the polynomial rows.  Read when a stream or bank is created / a stream call is made;
 *     set it before.  Default 0: the reference's behaviour.
 * SAVGOL_HIP_OPT_TILE_WIDTH: which of the two tile widths the 1-D batch kernels run with where both are built (fp32 half windows
 *   <= 18, fp64 <= 24).  0 (default): by job size -- the 12 / 16 KiB tile from 16384 tiles up, the 8 KiB tile below; 1: always the
 *   8 KiB tile; 2: always the wide one.  Same bits per output either way; a tuning and test knob.                                */
int         savgol_hip_set_option(int option, int value);
/* The same switches PER CALL: the *_ex forms of the 1-D device entry points take them as flags, so two threads (or two
 * calls) can run different summation orders or tile widths at the same time; the options above are only the DEFAULTS the
 * non-_ex entry points use (savgol_hip_default_flags() returns them as flags).  A flag word is complete: an _ex call
 * ignores the process-wide options.  TILE_NARROW and TILE_WIDE exclude each other; neither = by job size.               */
enum { SAVGOL_BATCH_REFERENCE_SUMMATION = 1u, SAVGOL_BATCH_PLAIN_SUMMATION = 2u, SAVGOL_BATCH_TILE_NARROW = 4u, SAVGOL_BATCH_TILE_WIDE = 8u,
       SAVGOL_BATCH_CORRECT_LEADING_EDGE = 16u, SAVGOL_BATCH_BOUNDARY_AWARE = 32u /* strided calls only */,
       SAVGOL_BATCH_MOMENT_F64 = 64u /* fp64 calls, half_window 24..32: block moments replace the taps on the lanes' common block (~36 multiply-adds
                                        per output instead of 65).  The block's share comes from the polynomial fitted to the fp32 table, so the
                                        result is within ~1e-7 (bar: 1e-6) of the default fp64 path instead of its 1e-12.  Opt-in for that reason. */ };
unsigned    savgol_hip_default_flags(void);
/* Diagnostic (host only, no device needed): the constant table the half-lane block-moment kernel (fp32 batch calls, half windows 20..32:
 * csrc/sg_k1d_momenth.hpp; layout in csrc/sg_k1d_host.hpp, at most SAVGOL_HIP_MOMENT_TABLE_FLOATS floats).  Returns the number of block
 * moments the kernel will use (3, 5 or 7), 0 when the filter keeps the plain 2n+1-tap sum (half windows below 20, poly_order > 6, tables that
 * are not a polynomial), -1 on NULL.                                                                                                       */
#define SAVGOL_HIP_MOMENT_TABLE_FLOATS 400
int         savgol_hip_momenth_table(const SavgolFilter *filter, float *table);
/* Diagnostic (host only): the fit behind the fused stream bank's block-moment tiles (csrc/sg_stream_dma.hip, half windows 12..20).  `center_weights`:
 * the 2n+1 fp32 taps a bank applies; `coefficients`: 3 x SAVGOL_HIP_STREAM_MOMENT_OFFSETS floats, [s][off] = weight of moment s (basis 1, t - 3.5,
 * (t - 3.5)^2 - 5.25 on t = 0..7) of the 8-tick block that starts `off` taps into a window.  Returns the number of moments (1..3), 0 when the
 * table is not a polynomial of degree <= 2 to 3e-7 of its largest tap (the bank then runs tap by tap), -1 on NULL.                              */
#define SAVGOL_HIP_STREAM_MOMENT_OFFSETS 34
int         savgol_hip_stream_moment_table(int half_window, const float *center_weights, float *coefficients);

/* ---------------------------------------------------------------- table export ------- *
 * The reference's on-disk format for a filter's tables: the C header its savgol_export tool writes
 * (src/savgol_export.c:145-282 -- `%+.10ef` values four per line, CENTER_WEIGHTS[2n+1], EDGE_WEIGHTS[n][2n+1], a naive
 * inline apply).  Host only.  `prefix` NULL = "SAVGOL"; `timestamp` NULL = now ("%Y-%m-%d %H:%M:%S", the tool's
 * "Generated by" line).  Writes at most capacity-1 bytes + NUL into buf (buf may be NULL) and returns the full length
 * of the text, -1 on a bad filter.  Byte-identical to the tool's output for the same filter, prefix and timestamp.   */
long savgol_export_header(const SavgolFilter *filter, const char *prefix, const char *timestamp, char *buf, size_t capacity);

/* ---------------------------------------------------------------- 1-D batch ----------- *
 * channels independent signals, row-major: sample i of channel c at base[c*ld + i].
 * Arithmetic of savgol_apply (reference src/savgolFilter.c:743-804): centre taps on the
 * interior, filter->config.boundary on the first/last n samples (POLYNOMIAL rows incl. the
 * reference's reversed leading edge; REFLECT / PERIODIC / CONSTANT by index remap).
 * No row of d_in may share a byte with a row of d_out (checked: -1); interleaved layouts whose rows stay apart are fine
 * (in = buf[:, 0, :], out = buf[:, 1, :] with equal pitches).  IN PLACE is the one exception, and its contract is exact: d_out == d_in AND
 * out_ld == in_ld, the full-length entry points only (not the _valid ones), channels of at most 2^30 samples.  The call then returns the
 * OUT-OF-PLACE answer (bit-identical to it; the reference's own in-place loop reads samples it has already overwritten -- a documented
 * divergence, as for the host-pointer call): the even tiles of every channel run first and hand their first / last samples to their odd
 * neighbours through a stream-ordered stash of 2n samples per tile, then the odd tiles, then the edge rows (+3-6 % over out of place).  length >= 2n+1, and any size_t beyond that, like the
 * reference's: a channel longer than 2^30 samples is enqueued as sub-rows of 2^29 outputs plus its two
 * ends (same arithmetic per output; a little stream-ordered scratch for the ends).
 * f32: fp32 tables, fp32 FMA accumulation (within 1e-6 normwise of the fp64 oracle).
 * f64: the same fp32 tables promoted exactly to double, double accumulation, the reference's
 *      float 1/dt_scale promoted -- there is no fp64 path in the reference.
 *      The centre taps must be (anti)symmetric bit for bit, as savgol_create builds them
 *      (tap[k] == +-tap[2n-k]); a hand-edited table that is not returns -1.                   */
int savgol_apply_batch_f32(const SavgolFilter *filter, const float *d_in, float *d_out,
                           size_t channels, size_t length, size_t in_ld, size_t out_ld,
                           void *stream);
int savgol_apply_batch_f64(const SavgolFilter *filter, const double *d_in, double *d_out,
                           size_t channels, size_t length, size_t in_ld, size_t out_ld,
                           void *stream);
int savgol_apply_batch_f32_ex(const SavgolFilter *filter, const float *d_in, float *d_out,
                              size_t channels, size_t length, size_t in_ld, size_t out_ld,
                              unsigned flags, void *stream);
int savgol_apply_batch_f64_ex(const SavgolFilter *filter, const double *d_in, double *d_out,
                              size_t channels, size_t length, size_t in_ld, size_t out_ld,
                              unsigned flags, void *stream);
/* fp64 with the tolerance stated in the call: `rel_tol` is the normwise distance from the fp64 oracle (promoted fp32 tables, double accumulation)
 * the caller accepts, and it picks the kernel.  rel_tol >= 1e-6 -- the bar BASELINE's fp64 configs state -- runs the block-moment kernel at half
 * windows 24..32 when the filter's table is the polynomial savgol_create builds (measured <= 1.5e-7 of the oracle, 0.78-0.84 of the HBM roofline at
 * n = 32 against the default's 0.66-0.70); a tighter rel_tol, other half windows and hand-edited tables run the tap-by-tap kernel (1e-12).  Same
 * process-wide defaults as the plain entry points otherwise.  The _ex flag SAVGOL_BATCH_MOMENT_F64 is the same choice as a flag.           */
int savgol_apply_batch_f64_tol(const SavgolFilter *filter, const double *d_in, double *d_out,
                               size_t channels, size_t length, size_t in_ld, size_t out_ld,
                               double rel_tol, void *stream);
int savgol_apply_valid_batch_f64_tol(const SavgolFilter *filter, const double *d_in, double *d_out,
                                     size_t channels, size_t length, size_t in_ld, size_t out_ld,
                                     double rel_tol, void *stream);
/* savgol_apply_valid (:821-850) per channel: writes length-2n samples at d_out[c*out_ld + 0..] */
int savgol_apply_valid_batch_f32(const SavgolFilter *filter, const float *d_in, float *d_out,
                                 size_t channels, size_t length, size_t in_ld, size_t out_ld,
                                 void *stream);
int savgol_apply_valid_batch_f64(const SavgolFilter *filter, const double *d_in, double *d_out,
                                 size_t channels, size_t length, size_t in_ld, size_t out_ld,
                                 void *stream);
int savgol_apply_valid_batch_f32_ex(const SavgolFilter *filter, const float *d_in, float *d_out,
                                    size_t channels, size_t length, size_t in_ld, size_t out_ld,
                                    unsigned flags, void *stream);
int savgol_apply_valid_batch_f64_ex(const SavgolFilter *filter, const double *d_in, double *d_out,
                                    size_t channels, size_t length, size_t in_ld, size_t out_ld,
                                    unsigned flags, void *stream);
/* savgol_apply_strided (:877-934) on device memory: element i of channel c is the float at
 * (char*)base + c*channel_pitch + i*stride + offset (bytes).  Edges are always POLYNOMIAL.     */
int savgol_apply_strided_batch_f32(const SavgolFilter *filter,
                                   const void *d_in, size_t in_stride, size_t in_offset, size_t in_channel_pitch,
                                   void *d_out, size_t out_stride, size_t out_offset, size_t out_channel_pitch,
                                   size_t channels, size_t count, void *stream);
int savgol_apply_strided_batch_f32_ex(const SavgolFilter *filter,
                                      const void *d_in, size_t in_stride, size_t in_offset, size_t in_channel_pitch,
                                      void *d_out, size_t out_stride, size_t out_offset, size_t out_channel_pitch,
                                      size_t channels, size_t count, unsigned flags, void *stream);

/* ---------------------------------------------------------------- stream bank --------- *
 * `streams` independent SavgolStream-equivalents advancing in lock step, state in HBM as a
 * structure of arrays (ring[slot][stream]).  Arithmetic of src/savgol_stream.c: single fp32
 * accumulator, taps in order, separate multiply and add -- outputs are bit-identical to the
 * reference's savgol_stream_push / _push_full / _flush / _flush_leading for every stream.
 * Output rows are [row][stream] with pitch `streams`.                                        */
typedef struct SavgolStreamBank SavgolStreamBank;

SavgolStreamBank *savgol_streambank_create(const SavgolConfig *config, size_t streams);
/* Per-bank choice of the summation, fixed at create (the stream analogue of the 1-D batch path's fast default):
 * SAVGOL_STREAMBANK_FMA: centre outputs of _push / _push_block / _push_full (after the filling tick) are fused multiply-adds --
 *   one v_pk_fma_f32 per tap and stream pair instead of a multiply and an add (reference loop src/savgol_stream.c:25-38), half
 *   the vector instructions of the block push.  NOT the reference's bits: <= 1e-6 (smoothing) / 1.5e-6 (derivatives) normwise of
 *   the fp64 oracle, like the default 1-D batch kernels.  Edge rows (leading burst, _flush, _flush_leading), the resident
 *   tick service and calls of >= 2^31 ticks keep the reference's order.  Half windows <= 16 sum in two interleaved chains in _push
 *   and _push_block alike; above 16 the block push keeps ONE chain per output (its accumulators live in registers) while the
 *   per-tick kernel uses two, so the two calls agree to fp32 rounding there, not bit for bit.  flags 0 == savgol_streambank_create. */
enum { SAVGOL_STREAMBANK_FMA = 1 };
SavgolStreamBank *savgol_streambank_create_ex(const SavgolConfig *config, size_t streams, unsigned flags);
void   savgol_streambank_destroy(SavgolStreamBank *bank);
int    savgol_streambank_reset(SavgolStreamBank *bank, void *stream);
/* one tick = one sample per stream.  Returns 1 when d_out[0..streams) holds centre outputs,
 * 0 while the windows are still filling (d_out untouched), -1 on error.                       */
int    savgol_streambank_push(SavgolStreamBank *bank, const float *d_samples, float *d_out, void *stream);
/* The same tick, and d_out is complete (in memory: write-through stores) when the call returns: the tick kernel's last block writes a sequence
 * number into a word of pinned host memory and the host spins on it -- no hipStreamSynchronize.  The lowest-latency way to take one tick's
 * outputs.  Needs a multiple of 64 streams; otherwise it is push + hipStreamSynchronize.  Same return value as savgol_streambank_push. */
int    savgol_streambank_push_wait(SavgolStreamBank *bank, const float *d_samples, float *d_out, void *stream);
/* with edges: returns the number of output rows written (0, 1, or up to n+1 on the tick that
 * fills the windows, truncated to max_rows), -1 on error.                                     */
int    savgol_streambank_push_full(SavgolStreamBank *bank, const float *d_samples,
                                   float *d_out, int max_rows, void *stream);
/* `ticks` pushes in one launch (ring kept on chip in between): d_samples[t*streams + s],
 * d_out[t*streams + s] is written for every tick t that has a centre output.  Returns the
 * number of ticks that produced output (they are the last ones), -1 on error.                 */
int    savgol_streambank_push_block(SavgolStreamBank *bank, const float *d_samples, size_t ticks,
                                    float *d_out, void *stream);
/* trailing / leading edge rows, up to n of them; -1 on bad arguments, 0 if never filled       */
int    savgol_streambank_flush(SavgolStreamBank *bank, float *d_out, int max_rows, void *stream);
int    savgol_streambank_flush_leading(SavgolStreamBank *bank, float *d_out, int max_rows, void *stream);
bool   savgol_streambank_ready(const SavgolStreamBank *bank);
size_t savgol_streambank_latency(const SavgolStreamBank *bank);
size_t savgol_streambank_streams(const SavgolStreamBank *bank);
size_t savgol_streambank_samples_received(const SavgolStreamBank *bank);   /* per stream */
size_t savgol_streambank_samples_output(const SavgolStreamBank *bank);     /* per stream */
/* Resident tick service: ONE kernel stays on the device (one wave per 256 streams, accumulators in registers) and every tick
 * is a doorbell write + a spin on a completion array instead of a launch + a synchronise (csrc/sg_stream_service.hip).
 * service_tick is SYNCHRONOUS: it returns 1 when d_out[0..streams) holds this tick's centre outputs (complete in device
 * memory), 0 while the windows are still filling (d_out untouched), -1 on error; results are bit-identical to
 * savgol_streambank_push / the reference's savgol_stream_push.  Needs streams % 4 == 0, streams <= 262144, 16-byte aligned
 * d_samples / d_out.  While the service runs, the other savgol_streambank_* calls on this bank return -1 (stop first; the
 * ring, write position and counters carry over in both directions).  The kernel leaves by itself after idle_ms (0 = 1000)
 * without a tick and is restarted by the next one; until then a DEVICE-wide synchronise waits for it -- synchronise
 * streams, or stop the service.                                                                                     */
int    savgol_streambank_service_start(SavgolStreamBank *bank, unsigned idle_ms);
int    savgol_streambank_service_tick(SavgolStreamBank *bank, const float *d_samples, float *d_out);
int    savgol_streambank_service_stop(SavgolStreamBank *bank);
int    savgol_streambank_service_running(const SavgolStreamBank *bank);
/* checkpoint / resume: the whole state as one blob (synchronous) */
size_t savgol_streambank_state_bytes(const SavgolStreamBank *bank);
int    savgol_streambank_save(const SavgolStreamBank *bank, void *host_blob, void *stream);
int    savgol_streambank_load(SavgolStreamBank *bank, const void *host_blob, void *stream);

/* ---------------------------------------------------------------- 2-D batch ----------- *
 * `images` frames, image k at base + k*image_pitch (elements), row pitch in elements.
 * Arithmetic of savgol2d_apply / savgol2d_apply_valid (src/savgol2d.c:356-456).
 * method: 0 = auto, 1 = direct dense window (bit-identical to the reference), 2 = exact low-rank
 * separable passes (rolling-window kernel for every half window: one launch holds 4 terms to n = 8, 3 to n = 12, 2 to
 * n = 16; kernels with more -- orders 4 to 6 on wide windows -- take two launches, the second adding to the output frame,
 * which therefore must not overlap the input), 3 = the separable tile kernel for any half window (diagnostic).
 * Rectangular windows (half_window_x != half_window_y): methods 0 / 2 run the rolling kernel of the LARGER half window on
 * factors zero-padded to it (same results to fp32 rounding).  Caveat of the padding: a zero tap times a non-finite sample is NaN, so in
 * this method NaN / Inf input spreads over the padded (square) window instead of the rectangular one; method 1 does not.
 * The input and output frame stacks must not share a byte (no 2-D kernel can run in place: every output reads its neighbours'
 * inputs): an overlapping call returns -1 with "overlap" in savgol_hip_last_error(); the derivative calls below alike.        */
int savgol2d_apply_batch_f32(const Savgol2DFilter *filter,
                             const float *d_in, int rows, int cols, int in_stride, size_t in_image_pitch,
                             float *d_out, int out_stride, size_t out_image_pitch,
                             size_t images, Savgol2DBoundary boundary, int method, void *stream);

/* Derivative frames (arithmetic of savgol2d_gradient / _hessian / _laplacian, src/savgol2d.c:462-618).  The Laplacian
 * uses the single summed kernel (scale_xx*Wxx + scale_yy*Wyy): no temporary frame, no add pass.  Outputs may be NULL
 * (skipped), like the reference.  Square windows use the separable kernels (fp32 rounding only vs the reference):
 * three Hessian frames, and frames whose rank the rolling-window kernel is not built for at that half window, come from
 * ONE read of each input tile (tile kernel); one or two frames otherwise are rolling-window launches (the gradient of
 * a half window <= 8: one launch for both frames; faster).  Other window shapes: one dense pass per
 * output.                                                                                                           */
int savgol2d_gradient_batch_f32(int half_win_x, int half_win_y, int poly_order,
                                const float *d_in, int rows, int cols, int in_stride, size_t in_image_pitch,
                                float *d_grad_x, float *d_grad_y, int out_stride, size_t out_image_pitch,
                                size_t images, float delta_x, float delta_y, Savgol2DBoundary boundary, void *stream);
int savgol2d_hessian_batch_f32(int half_win_x, int half_win_y, int poly_order,
                               const float *d_in, int rows, int cols, int in_stride, size_t in_image_pitch,
                               float *d_xx, float *d_xy, float *d_yy, int out_stride, size_t out_image_pitch,
                               size_t images, float delta_x, float delta_y, Savgol2DBoundary boundary, void *stream);
int savgol2d_laplacian_batch_f32(int half_win_x, int half_win_y, int poly_order,
                                 const float *d_in, int rows, int cols, int in_stride, size_t in_image_pitch,
                                 float *d_out, int out_stride, size_t out_image_pitch,
                                 size_t images, float delta_x, float delta_y, Savgol2DBoundary boundary, void *stream);

/* ---------------------------------------------------------------- 2-D row bands (multi-GPU) --- *
 * Frames too large for one GPU, or fewer frames than GPUs: every rank owns a horizontal band of rows of every frame and needs
 * the half_window_y rows next to its band from the ranks above and below -- the only exchange step on the hot path (the
 * reference filters whole frames: src/savgol2d.c:398-456; a band's rows must be the rows that call would have produced).
 * savgol2d_rowband_plan: rank's rows [*row_lo, *row_hi) of a `rows`-row frame (savgol_hip_shard_range) and how many halo rows it
 *   needs from above / below (half_window_y, or 0 at the frame's real top / bottom).  -1 when the bands would be thinner than
 *   2 x half_window_y.
 * savgol2d_apply_rowband_f32: filters one band, d_band = its band_rows x cols rows per image.  d_halo_up / d_halo_down: the
 *   half_window_y rows just above / below the band (row 0 = the farthest-up row; halo_stride elements between rows,
 *   halo_image_pitch between images), NULL where the band ends at the frame's real edge -- the boundary mode applies there
 *   exactly as in savgol2d_apply_batch_f32.  All band_rows output rows are written (VALID: not the frame's own first / last
 *   half_window_y rows, not the half_window_x border columns).  Enqueue-only plus a stream-ordered scratch allocation; the
 *   FIRST launch (the band itself) does not read the halos.  Method 1 gives the whole-frame call's bits; method 2 its bits for
 *   kernels of order > 3 or derivative kernels, and fp32 rounding (<= 2.5e-7 of the input's maximum) for the additive smoothing
 *   kernels, whose rolling column sums are re-seeded on a phase tied to the frame's row 0.
 * The halo buffers are the caller's to fill: a device-to-device copy on one GPU, ncclSend / ncclRecv across GPUs --
 * savgol2d_rowband_exchange_rccl in the optional lib/libsavgol_hip_rccl.so (csrc/sg_rowband_rccl.cpp; comm = an ncclComm_t,
 * d_send_scratch = 2 * images * half_window_y * cols floats, halos received with halo_stride = cols and halo_image_pitch =
 * half_window_y * cols; one message per neighbour and direction for the whole stack).                                   */
int savgol2d_rowband_plan(int rows, int half_win_y, int rank, int world_size, int *row_lo, int *row_hi, int *halo_up, int *halo_down);
int savgol2d_apply_rowband_f32(const Savgol2DFilter *filter,
                               const float *d_band, int band_rows, int cols, int in_stride, size_t in_image_pitch,
                               const float *d_halo_up, const float *d_halo_down, int halo_stride, size_t halo_image_pitch,
                               float *d_out, int out_stride, size_t out_image_pitch,
                               size_t images, Savgol2DBoundary boundary, int method, void *stream);
/* Step (2) of the call above on its own -- the half_window_y output rows next to each artificial edge -- for callers that overlap
 * the halo exchange with the band: enqueue savgol2d_apply_batch_f32 on the band (it reads no halo) while the exchange runs on
 * another stream, make the compute stream wait for the exchange, then call this.  Same arguments, same result as the one call. */
int savgol2d_apply_rowband_edges_f32(const Savgol2DFilter *filter,
                                     const float *d_band, int band_rows, int cols, int in_stride, size_t in_image_pitch,
                                     const float *d_halo_up, const float *d_halo_down, int halo_stride, size_t halo_image_pitch,
                                     float *d_out, int out_stride, size_t out_image_pitch,
                                     size_t images, Savgol2DBoundary boundary, int method, void *stream);
/* The same on two streams (round 6): the strips are gathered and filtered on `halo_stream` -- the stream the halos arrive on; call this after
 * enqueueing the exchange there.  They read the band and the halos and write library scratch only, so they run BESIDE the band launch still
 * going on `stream`; only the copy of the finished half_window_y rows per edge into d_out is ordered behind everything enqueued on `stream`
 * so far (the band launch), through an event the call records itself.  No wait between the two streams is needed from the caller.
 * halo_stream == stream is savgol2d_apply_rowband_edges_f32. */
int savgol2d_apply_rowband_edges_streams_f32(const Savgol2DFilter *filter,
                                             const float *d_band, int band_rows, int cols, int in_stride, size_t in_image_pitch,
                                             const float *d_halo_up, const float *d_halo_down, int halo_stride, size_t halo_image_pitch,
                                             float *d_out, int out_stride, size_t out_image_pitch,
                                             size_t images, Savgol2DBoundary boundary, int method, void *halo_stream, void *stream);
/* (savgol2d_rowband_exchange_rccl is declared in savgol_hip_rccl.h -- that library is the only part that links librccl) */

/* ---------------------------------------------------------------- bench utilities ----- *
 * Synthetic workload of SURVEY.md section 8(d), generated in HBM (never crosses PCIe):
 * x[c][i] = sin(2 pi f_c i) + 0.5 sin(2 pi 7.3 f_c i + phi_c) + 0.1 u(c,i).                  */
int savgol_hip_synth_f32(float *d_dst, size_t channel0, size_t channels, size_t length, size_t ld,
                         uint64_t seed, void *stream);
int savgol_hip_synth_f64(double *d_dst, size_t channel0, size_t channels, size_t length, size_t ld,
                         uint64_t seed, void *stream);

/* What the memory system gives a plain stream of the same buffers, for the bench line's roofline.copy_frac / read_only_frac (SURVEY.md 8d:
 * "a device copy timed in the same harness"): one 16-byte nontemporal vector per thread.  bytes and both addresses multiples of 16.
 * _read stores nothing (d_sink4 may be NULL; 4 bytes that are never written in practice).  Enqueue only.  0 / -1. */
int savgol_hip_stream_copy(const void *d_in, void *d_out, size_t bytes, void *stream);
int savgol_hip_stream_read(const void *d_in, size_t bytes, void *d_sink4, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SAVGOL_HIP_H */
