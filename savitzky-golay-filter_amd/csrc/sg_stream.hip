// sg_stream.hip -- streaming path: the 13 drop-in entry points of savgol_stream.h (one stream, state
// in the caller-visible POD) and the stream bank of savgol_hip.h (many streams, state in HBM).
//
// Reference arithmetic (src/savgol_stream.c): every output is ONE ring-buffer dot product,
//   centre   : sum_i cw[i]    * ring[(wp + i)        % ws]     (convolve_center_circular :25-38)
//   trailing : sum_i ew[e][i] * ring[(wp + i)        % ws]     (convolve_edge_trailing   :43-56)
//   leading  : sum_i ew[e][i] * ring[(wp + ws-1 - i) % ws]     (convolve_edge_leading    :61-74)
// with a single fp32 accumulator, taps ascending, separate multiply and add, then * dt_inv.
// The kernels below keep exactly that order with __fmul_rn / __fadd_rn (no FMA contraction), so the
// outputs are bit-identical to the reference's -- this path is latency/launch bound, the extra
// rounding step costs nothing.
//
// Bank layout: ring[slot][stream] (stream fastest), one shared write position because all streams
// of a bank tick together.  A tick reads ws rows of `streams` floats (coalesced, L2/Infinity-Cache
// resident: 8.65 MB for 65 536 streams at n=16) and writes one row + one output row.
// savgol_streambank_push_block keeps each stream's ring in LDS for `ticks` pushes and touches HBM
// only for the samples and the outputs.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <immintrin.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "sg_internal.h"
#include "sg_pk.hpp"
#include "sg_runtime.hpp"
#include "sg_stream.hpp"

extern "C" int sg_small_stream_rows(void *ctx, const float *d_table, const float *ring, int ws, int wp, float dt_inv, int count, const int *row,
                                    const int *backward, float *output);

namespace sg {

// one ring dot product, reference order: taps ascending, ring walked forward from the oldest sample
// (slot wp) or backward from the newest (slot wp-1), wrapping once.  `w` points into LDS (broadcast
// reads).  The loads are issued 8 at a time so their latency overlaps; the multiply-add chain itself stays
// strictly sequential (that is what makes the result bit-identical to the reference).
template <typename Load>
__device__ __forceinline__ float ring_dot(Load load, const float *w, int ws, int wp, bool backward)
{
    float acc = 0.0f;
    int i = 0;
    for (; i + 8 <= ws; i += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int slot = backward ? (wp + ws - 1 - (i + j)) : (wp + i + j);
            if (slot >= ws) slot -= ws;
            v[j] = load(slot);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __fadd_rn(acc, __fmul_rn(w[i + j], v[j]));
    }
    for (; i < ws; ++i) {
        int slot = backward ? (wp + ws - 1 - i) : (wp + i);
        if (slot >= ws) slot -= ws;
        acc = __fadd_rn(acc, __fmul_rn(w[i], load(slot)));
    }
    return acc;
}

__device__ __forceinline__ float ring_dot_global(const float *__restrict__ ring, size_t streams, size_t s,
                                                 const float *w, int ws, int wp, bool backward)
{
    return ring_dot([&](int slot) { return ring[(size_t)slot * streams + s]; }, w, ws, wp, backward);
}

// savgol_streambank_push_wait: the tick kernel itself tells the host that it is done.  Outputs leave with write-through stores; every wave
// drains them (vmcnt(0)), the block meets, one lane takes a ticket; the block that draws the last ticket re-arms the counter and writes the
// sequence number into a word of pinned host memory the caller spins on -- no hipStreamSynchronize (5-6 us) on the way.
struct TickSignal {
    unsigned *counter;               // device word, zero between ticks
    unsigned *flag;                  // device view of a pinned host word
    unsigned  seq;                   // what the host waits for; counter == nullptr: no signalling
};
__device__ __forceinline__ void tick_signal(const TickSignal &sig)
{
    if (!sig.counter) return;                                      // uniform
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(sig.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == gridDim.x - 1) {
            __hip_atomic_store(sig.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sig.flag, sig.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__device__ __forceinline__ void store_out(float *p, float v, bool through)
{
    if (through) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // write-through: in memory once vmcnt drains
    else *p = v;
}

// write one sample per stream at slot wp_old, then (if `emit`) the centre output of the window that
// now starts at wp_new = (wp_old + 1) % ws
__global__ __launch_bounds__(256) void sg_bank_tick_kernel(float *__restrict__ ring, const float *__restrict__ samples,
                                                           float *__restrict__ out, size_t streams,
                                                           const float *__restrict__ table, int ws, int wp_old,
                                                           float dt_inv, int emit, const TickSignal sig)
{
    __shared__ float wl[SAVGOL_MAX_WINDOW];
    if (emit) {
        for (int i = threadIdx.x; i < ws; i += blockDim.x) wl[i] = table[i];
        __syncthreads();
    }
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < streams) {                                             // (no early return: every thread reaches tick_signal's barrier exactly once)
        ring[(size_t)wp_old * streams + s] = samples[s];
        if (emit) {
            int wp = wp_old + 1;
            if (wp >= ws) wp -= ws;
            // the slot just written is read back by the same thread: program order is enough
            store_out(out + s, __fmul_rn(ring_dot_global(ring, streams, s, wl, ws, wp, false), dt_inv), sig.counter != nullptr);
        }
    }
    tick_signal(sig);
}

// rows of outputs from the current ring contents: row r uses table row rows[r] (0 = centre,
// 1+e = edge row e), walked backward (leading edge) or forward
struct RowList { int count; int row[SAVGOL_MAX_HALF_WINDOW + 1]; int backward[SAVGOL_MAX_HALF_WINDOW + 1]; };

__global__ __launch_bounds__(256) void sg_bank_rows_kernel(const float *__restrict__ ring, float *__restrict__ out,
                                                           size_t streams, const float *__restrict__ table, int ws,
                                                           int wp, float dt_inv, const RowList rows)
{
    __shared__ float wl[SAVGOL_MAX_WINDOW];
    const int r = blockIdx.y;
    for (int i = threadIdx.x; i < ws; i += blockDim.x) wl[i] = table[(size_t)rows.row[r] * ws + i];
    __syncthreads();
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= streams) return;
    out[(size_t)r * streams + s] = __fmul_rn(ring_dot_global(ring, streams, s, wl, ws, wp, rows.backward[r] != 0), dt_inv);
}

// `ticks` pushes in one launch.  Tick t's centre output is a dot product over the last 2n+1 samples of the
// sequence "ring contents (oldest first), then this call's samples": the block push is a convolution down the
// time axis of a [tick][stream] array.  So it is tiled like one: a block = 64 streams x 64 ticks; the 64+2n rows
// it needs (rows before this call come out of the ring) go to LDS with coalesced 256-B row reads; a lane owns one
// stream and 16 consecutive ticks and walks its 16+2n rows once, feeding each into the accumulators it touches.
// Every accumulator still sees its taps in ascending order with separate multiply and add -> bit-identical to the
// per-tick kernel and to the reference; 16 independent chains per lane hide the add latency.  HBM traffic = the
// samples and the outputs (8 B/sample); afterwards sg_bank_store_tail_kernel writes the newest 2n+1 samples back.
struct alignas(8) StreamTaps { float w[SAVGOL_MAX_WINDOW + 1]; };     // by-value kernarg -> 33 aligned SGPR pairs

// newest min(ticks, WS) samples of the call -> their ring slots (sample q of the call lands in slot (wp0 + q) mod WS)
__global__ __launch_bounds__(256) void sg_bank_store_tail_kernel(float *__restrict__ ring, const float *__restrict__ samples,
                                                                 size_t streams, int ws, int wp0, size_t ticks)
{
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= streams) return;
    const size_t first = ticks > (size_t)ws ? ticks - (size_t)ws : 0;
    for (size_t q = first + blockIdx.y; q < ticks; q += gridDim.y)
        ring[(size_t)((wp0 + q) % (size_t)ws) * streams + s] = samples[q * streams + s];
}

// One tick with every load in flight at once: half window known at compile time, so the 2n ring rows a stream needs
// are 2n independent loads (the newest sample comes straight from `samples`), then the strictly ordered multiply-add
// chain.  The generic kernel above waits for the ring in batches of 8.
template <int N, bool FMA>
__global__ __launch_bounds__(256) void sg_bank_tick_n_kernel(float *__restrict__ ring, const float *__restrict__ samples,
                                                             float *__restrict__ out, size_t streams, const StreamTaps taps,
                                                             int wp_old, float dt_inv, const TickSignal sig)
{
    constexpr int WS = 2 * N + 1;
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= streams) {
        if (sig.counter) tick_signal(sig);                         // only whole-wave tails come here (256-thread blocks, streams a multiple of 64 when signalling)
        return;
    }
    const float xnew = samples[s];
    ring[(size_t)wp_old * streams + s] = xnew;
    float v[WS];
    int slot = wp_old + 1;                                  // oldest sample
    if (slot >= WS) slot = 0;
#pragma unroll
    for (int i = 0; i < WS - 1; ++i) {
        v[i] = ring[(size_t)slot * streams + s];
        if (++slot >= WS) slot = 0;
    }
    v[WS - 1] = xnew;
    float acc = 0.0f;
    if constexpr (FMA) {
        // SAVGOL_STREAMBANK_FMA: fused multiply-adds on two chains (even taps, odd taps), as the block-push kernel's fast form
        float odd = __fmul_rn(taps.w[1], v[1]);
        acc = __fmul_rn(taps.w[0], v[0]);
#pragma unroll
        for (int i = 2; i < WS; ++i) {
            if (i & 1) odd = __fmaf_rn(taps.w[i], v[i], odd);
            else       acc = __fmaf_rn(taps.w[i], v[i], acc);
        }
        acc = __fadd_rn(acc, odd);
    } else {
#pragma unroll
        for (int i = 0; i < WS; ++i) acc = __fadd_rn(acc, __fmul_rn(taps.w[i], v[i]));
    }
    store_out(out + s, __fmul_rn(acc, dt_inv), sig.counter != nullptr);
    tick_signal(sig);
}

template <int N>
static int dispatch_tick(int n, bool fma, float *ring, const float *samples, float *out, size_t streams, const StreamTaps &taps, int wp_old,
                         float dt_inv, hipStream_t st, const TickSignal &sig)
{
    if (n == N) {
        if (fma)
            hipLaunchKernelGGL((sg_bank_tick_n_kernel<N, true>), dim3((unsigned)((streams + 255) / 256)), dim3(256), 0, st, ring, samples, out,
                               streams, taps, wp_old, dt_inv, sig);
        else
            hipLaunchKernelGGL((sg_bank_tick_n_kernel<N, false>), dim3((unsigned)((streams + 255) / 256)), dim3(256), 0, st, ring, samples, out,
                               streams, taps, wp_old, dt_inv, sig);
        return 1;
    }
    if constexpr (N < SAVGOL_MAX_HALF_WINDOW) return dispatch_tick<N + 1>(n, fma, ring, samples, out, streams, taps, wp_old, dt_inv, st, sig);
    else return 0;
}

// single stream (host drop-in API): the ring sits in pinned host memory (the caller-visible POD is the
// state), results go back to pinned host memory; thread r computes output row r
__global__ __launch_bounds__(64) void sg_stream_rows_kernel(const float *__restrict__ ring, float *__restrict__ out,
                                                            const float *__restrict__ table, int ws, int wp,
                                                            float dt_inv, const RowList rows)
{
    __shared__ float r_lds[SAVGOL_MAX_WINDOW];
    for (int j = threadIdx.x; j < ws; j += 64) r_lds[j] = ring[j];
    __syncthreads();
    const int r = threadIdx.x;
    if (r >= rows.count) return;
    const float *w = table + (size_t)rows.row[r] * ws;
    float acc = 0.0f;
    for (int i = 0; i < ws; ++i) {
        int slot = rows.backward[r] ? (wp + ws - 1 - i) : (wp + i);
        if (slot >= ws) slot -= ws;
        acc = __fadd_rn(acc, __fmul_rn(w[i], r_lds[slot]));
    }
    out[r] = __fmul_rn(acc, dt_inv);
}

// Device table for a filter, ws floats per row.  Reference behaviour (and PERIODIC always): row 0 = centre taps, row 1+e =
// polynomial edge row e, used backwards on the first window and forwards on the last (src/savgol_stream.c:43-74).
// With SAVGOL_HIP_OPT_BOUNDARY_AWARE and a REFLECT / CONSTANT filter the edge outputs are the centre taps on the index-remapped
// window instead (get_padded_sample, src/savgolFilter.c:442-482); a remapped window is still a dot product with the 2n+1
// samples in the ring, so the same kernels serve it from "effective" rows:
//     leading  output i      : w'_i[j]  = sum of cw[k] over the taps k whose remapped index lands on sample j of the first window
//     trailing output L-1-e  : w''_e[j] = the same on the last window
// rows 1..n hold w'_i REVERSED (the kernels read leading rows backwards), rows n+1..2n hold w''_e.
static bool boundary_aware_edges(const SavgolFilter *f)
{
    return sg_option_boundary_aware() && (f->config.boundary == SAVGOL_BOUNDARY_REFLECT || f->config.boundary == SAVGOL_BOUNDARY_CONSTANT);
}
static int trailing_row_base(const SavgolFilter *f) { return boundary_aware_edges(f) ? 1 + f->config.half_window : 1; }

static const float *filter_table(DeviceCtx *ctx, const SavgolFilter *f)
{
    const int n = f->config.half_window, ws = f->window_size;
    float packed[(2 * SAVGOL_MAX_HALF_WINDOW + 1) * SAVGOL_MAX_WINDOW];
    memcpy(packed, f->center_weights, sizeof(float) * ws);
    if (!boundary_aware_edges(f)) {
        for (int e = 0; e < n; ++e) memcpy(packed + (size_t)(1 + e) * ws, f->edge_weights[e], sizeof(float) * ws);
        return ctx_table(ctx, packed, sizeof(float) * (size_t)(n + 1) * ws, 0x57000000u + (unsigned)n);
    }
    const bool reflect = f->config.boundary == SAVGOL_BOUNDARY_REFLECT;
    for (int e = 0; e < n; ++e) {
        double lead[SAVGOL_MAX_WINDOW] = {}, trail[SAVGOL_MAX_WINDOW] = {};
        for (int k = 0; k < ws; ++k) {
            int j = e - n + k;                                   // leading output e reads first-window sample e - n + k
            if (j < 0) j = reflect ? -j - 1 : 0;
            lead[j] += (double)f->center_weights[k];
            j = ws - 1 - e - n + k;                              // trailing output L-1-e reads last-window sample ws-1-e-n+k
            if (j >= ws) j = reflect ? 2 * ws - j - 1 : ws - 1;
            trail[j] += (double)f->center_weights[k];
        }
        for (int j = 0; j < ws; ++j) {
            packed[(size_t)(1 + e) * ws + j] = (float)lead[ws - 1 - j];
            packed[(size_t)(1 + n + e) * ws + j] = (float)trail[j];
        }
    }
    return ctx_table(ctx, packed, sizeof(float) * (size_t)(2 * n + 1) * ws, 0x58000000u + (unsigned)n * 4u + (unsigned)f->config.boundary);
}

static bool filter_ok(const SavgolFilter *f)
{
    const int n = f->config.half_window;
    return n >= 1 && n <= SAVGOL_MAX_HALF_WINDOW && f->window_size == 2 * n + 1;
}

static inline float dt_inverse(const SavgolFilter *f) { return (f->dt_scale != 0.0f) ? (1.0f / f->dt_scale) : 1.0f; }

// run `rows` on the ring of a host-side stream; results copied into `dst`
static int single_stream_rows(const SavgolStream *st, const RowList &rows, float *dst)
{
    const SavgolFilter *f = st->filter;
    if (!filter_ok(f)) { sg_set_error("savgol_stream: filter struct is not a valid SavgolFilter"); return -1; }
    DeviceCtx *ctx = ctx_get();
    if (!ctx) return -1;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    const float *table = filter_table(ctx, f);
    if (!table) return -1;
    // the resident small-call service (sg_k1d_misc.hip): a doorbell instead of a launch + synchronise per sample -- same arithmetic
    if (sg_small_stream_rows(ctx, table, st->buffer, f->window_size, st->write_pos, st->dt_inv, rows.count, rows.row, rows.backward, dst) == 0) return 0;
    float *pinned = static_cast<float *>(ctx_pinned(ctx, sizeof(float) * 256));
    if (!pinned) return -1;
    float *ring = pinned + 64;
    memcpy(ring, st->buffer, sizeof(float) * SAVGOL_MAX_WINDOW);
    hipLaunchKernelGGL(sg_stream_rows_kernel, dim3(1), dim3(64), 0, nullptr, ring, pinned, table, f->window_size,
                       st->write_pos, st->dt_inv, rows);
    if (!hip_ok(hipGetLastError(), "stream kernel launch")) return -1;
    if (!hip_ok(hipStreamSynchronize(nullptr), "stream kernel")) return -1;
    memcpy(dst, pinned, sizeof(float) * rows.count);
    return 0;
}

}  // namespace sg

// =================================================================================================
// drop-in single-stream API (reference src/savgol_stream.c:80-315)
// =================================================================================================
extern "C" {

SavgolStream *savgol_stream_create(const SavgolConfig *config)
{
    if (!config) return nullptr;
    SavgolFilter *filter = savgol_create(config);
    if (!filter) return nullptr;
    SavgolStream *s = static_cast<SavgolStream *>(malloc(sizeof(SavgolStream)));
    if (!s) { savgol_destroy(filter); return nullptr; }
    s->filter = filter;
    s->owns_filter = true;
    s->dt_inv = sg::dt_inverse(filter);
    savgol_stream_reset(s);
    return s;
}

int savgol_stream_init(SavgolStream *stream, const SavgolFilter *filter)
{
    if (!stream || !filter) return -1;
    stream->filter = filter;
    stream->owns_filter = false;
    stream->dt_inv = sg::dt_inverse(filter);
    savgol_stream_reset(stream);
    return 0;
}

void savgol_stream_destroy(SavgolStream *stream)
{
    if (!stream) return;
    if (stream->owns_filter && stream->filter) savgol_destroy(const_cast<SavgolFilter *>(stream->filter));
    free(stream);
}

void savgol_stream_reset(SavgolStream *stream)
{
    if (!stream) return;
    stream->write_pos = 0;
    stream->samples_received = 0;
    stream->samples_output = 0;
    memset(stream->buffer, 0, sizeof(stream->buffer));
}

static void ring_store(SavgolStream *s, float sample)          // :162-164
{
    const int ws = s->filter->window_size;
    s->buffer[s->write_pos] = sample;
    s->write_pos = (s->write_pos + 1) % ws;
    s->samples_received++;
}

float savgol_stream_push(SavgolStream *stream, float sample, bool *output_valid)
{
    if (!stream || !stream->filter) { if (output_valid) *output_valid = false; return 0.0f; }
    ring_store(stream, sample);
    if (stream->samples_received < (size_t)stream->filter->window_size) {
        if (output_valid) *output_valid = false;
        return 0.0f;
    }
    sg::RowList rows; memset(&rows, 0, sizeof(rows));
    rows.count = 1;                                              // centre, forward
    float y = 0.0f;
    if (sg::single_stream_rows(stream, rows, &y) != 0) {
        fprintf(stderr, "savgol_stream_push: %s\n", savgol_hip_last_error());
        if (output_valid) *output_valid = false;
        return 0.0f;
    }
    stream->samples_output++;
    if (output_valid) *output_valid = true;
    return y;
}

int savgol_stream_push_full(SavgolStream *stream, float sample, float *output, int max_outputs)
{
    if (!stream || !stream->filter || !output || max_outputs <= 0) return 0;
    const int ws = stream->filter->window_size, n = stream->filter->config.half_window;
    const bool was_filling = stream->samples_received < (size_t)ws;
    ring_store(stream, sample);
    if (stream->samples_received < (size_t)ws) return 0;
    sg::RowList rows; memset(&rows, 0, sizeof(rows));
    if (was_filling) {                                           // n leading rows, then the centre (:205-221)
        for (int e = 0; e < n && rows.count < max_outputs; ++e) { rows.row[rows.count] = 1 + e; rows.backward[rows.count] = 1; rows.count++; }
        if (rows.count < max_outputs) { rows.row[rows.count] = 0; rows.backward[rows.count] = 0; rows.count++; }
    } else {
        rows.count = 1;
    }
    float tmp[SAVGOL_MAX_HALF_WINDOW + 1];
    if (sg::single_stream_rows(stream, rows, tmp) != 0) {
        fprintf(stderr, "savgol_stream_push_full: %s\n", savgol_hip_last_error());
        return 0;
    }
    memcpy(output, tmp, sizeof(float) * rows.count);
    stream->samples_output += rows.count;
    return rows.count;
}

int savgol_stream_flush(SavgolStream *stream, float *output, int max_count)
{
    if (!stream || !output || max_count <= 0) return -1;
    const SavgolFilter *f = stream->filter;
    const int n = f->config.half_window;
    if (stream->samples_received < (size_t)f->window_size) return 0;
    sg::RowList rows; memset(&rows, 0, sizeof(rows));
    rows.count = max_count < n ? max_count : n;
    for (int i = 0; i < rows.count; ++i) { rows.row[i] = sg::trailing_row_base(f) + (n - 1 - i); rows.backward[i] = 0; }   // :245-249
    float tmp[SAVGOL_MAX_HALF_WINDOW + 1];
    if (sg::single_stream_rows(stream, rows, tmp) != 0) {
        fprintf(stderr, "savgol_stream_flush: %s\n", savgol_hip_last_error());
        return -1;
    }
    memcpy(output, tmp, sizeof(float) * rows.count);
    stream->samples_output += rows.count;
    return rows.count;
}

int savgol_stream_flush_leading(SavgolStream *stream, float *output, int max_count)
{
    if (!stream || !output || max_count <= 0) return 0;
    const SavgolFilter *f = stream->filter;
    const int n = f->config.half_window;
    if (stream->samples_received < (size_t)f->window_size) return 0;
    sg::RowList rows; memset(&rows, 0, sizeof(rows));
    rows.count = max_count < n ? max_count : n;
    for (int i = 0; i < rows.count; ++i) { rows.row[i] = 1 + i; rows.backward[i] = 1; }             // :269-272
    float tmp[SAVGOL_MAX_HALF_WINDOW + 1];
    if (sg::single_stream_rows(stream, rows, tmp) != 0) {
        fprintf(stderr, "savgol_stream_flush_leading: %s\n", savgol_hip_last_error());
        return 0;
    }
    memcpy(output, tmp, sizeof(float) * rows.count);
    stream->samples_output += rows.count;
    return rows.count;
}

bool savgol_stream_ready(const SavgolStream *stream)
{
    return stream && stream->filter && stream->samples_received >= (size_t)stream->filter->window_size;
}

size_t savgol_stream_latency(const SavgolStream *stream)
{
    return (stream && stream->filter) ? stream->filter->config.half_window : 0;
}

size_t savgol_stream_buffered(const SavgolStream *stream)
{
    if (!stream || !stream->filter) return 0;
    const size_t ws = (size_t)stream->filter->window_size;
    return stream->samples_received < ws ? stream->samples_received : ws;
}

size_t savgol_stream_samples_received(const SavgolStream *stream) { return stream ? stream->samples_received : 0; }
size_t savgol_stream_samples_output(const SavgolStream *stream) { return stream ? stream->samples_output : 0; }

}  // extern "C"

// =================================================================================================
// stream bank (savgol_hip.h)
// =================================================================================================
namespace sg {
static unsigned bank_blocks(const SavgolStreamBank *b) { return (unsigned)((b->streams + 255) / 256); }
// A bank's ring and tables live on the device it was created on; launching from a thread whose current device is another
// one would hand foreign pointers to that GPU.  Every entry point that touches the device checks.
static bool bank_on_current_device(const SavgolStreamBank *b, const char *who)
{
    if (b->service) {                                      // the resident kernel owns the accumulators; the ring it keeps current
        sg_set_error("%s: the tick service is running on this bank -- savgol_streambank_service_stop() first", who);      // is only visible to other kernels after it has left
        return false;
    }
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != b->device) {
        sg_set_error("%s: the bank lives on device %d but this thread's current device is %d (call savgol_hip_set_device(%d) first)", who,
                     b->device, cur, b->device);
        return false;
    }
    return true;
}
}

extern "C" {

SavgolStreamBank *savgol_streambank_create(const SavgolConfig *config, size_t streams)
{
    return savgol_streambank_create_ex(config, streams, 0u);
}

SavgolStreamBank *savgol_streambank_create_ex(const SavgolConfig *config, size_t streams, unsigned flags)
{
    if (!config || streams == 0) { sg_set_error("savgol_streambank_create: bad arguments"); return nullptr; }
    if (flags & ~(unsigned)SAVGOL_STREAMBANK_FMA) { sg_set_error("savgol_streambank_create_ex: unknown flags 0x%x", flags); return nullptr; }
    SavgolFilter *f = savgol_create(config);
    if (!f) { sg_set_error("savgol_streambank_create: invalid configuration"); return nullptr; }
    sg::DeviceCtx *ctx = sg::ctx_get();
    if (!ctx) { savgol_destroy(f); return nullptr; }
    SavgolStreamBank *b = static_cast<SavgolStreamBank *>(calloc(1, sizeof(SavgolStreamBank)));
    if (!b) { savgol_destroy(f); return nullptr; }
    b->filter = f;
    b->streams = streams;
    b->device = ctx->ordinal;
    b->dt_inv = sg::dt_inverse(f);
    b->flags = flags;
    b->d_table = sg::filter_table(ctx, f);
    b->trail_base = sg::trailing_row_base(f);             // the option is read once, here
    const size_t bytes = sizeof(float) * (size_t)f->window_size * streams;
    if (!b->d_table || !sg::hip_ok(hipMalloc(reinterpret_cast<void **>(&b->d_ring), bytes), "hipMalloc(stream bank)") ||
        !sg::hip_ok(hipMemset(b->d_ring, 0, bytes), "hipMemset(stream bank)")) {
        if (b->d_ring) (void)hipFree(b->d_ring);
        savgol_destroy(f);
        free(b);
        return nullptr;
    }
    return b;
}

void savgol_streambank_destroy(SavgolStreamBank *bank)
{
    if (!bank) return;
    if (bank->service) (void)savgol_streambank_service_stop(bank);
    if (bank->d_ring) (void)hipFree(bank->d_ring);
    if (bank->signal) (void)hipHostFree(const_cast<unsigned *>(bank->signal));
    if (bank->signal_counter) (void)hipFree(bank->signal_counter);
    savgol_destroy(bank->filter);
    free(bank);
}

int savgol_streambank_reset(SavgolStreamBank *bank, void *stream)
{
    if (!bank) { sg_set_error("savgol_streambank_reset: NULL bank"); return -1; }
    if (!sg::bank_on_current_device(bank, "savgol_streambank_reset")) return -1;
    bank->wp = 0; bank->received = 0; bank->emitted = 0;
    const size_t bytes = sizeof(float) * (size_t)bank->filter->window_size * bank->streams;
    return sg::hip_ok(hipMemsetAsync(bank->d_ring, 0, bytes, static_cast<hipStream_t>(stream)), "hipMemsetAsync") ? 0 : -1;
}

static int bank_tick(SavgolStreamBank *bank, const float *d_samples, float *d_out, void *stream, const sg::TickSignal &sig, const char *who)
{
    if (!bank || !d_samples || !d_out) { sg_set_error("%s: NULL pointer", who); return -1; }
    if (!sg::bank_on_current_device(bank, who)) return -1;
    const int ws = bank->filter->window_size;
    const int emit = (bank->received + 1 >= (unsigned long long)ws) ? 1 : 0;
    if (emit) {
        sg::StreamTaps taps;
        memset(&taps, 0, sizeof(taps));
        memcpy(taps.w, bank->filter->center_weights, sizeof(float) * ws);
        sg::dispatch_tick<1>(bank->filter->config.half_window, (bank->flags & SAVGOL_STREAMBANK_FMA) != 0, bank->d_ring, d_samples, d_out, bank->streams, taps, bank->wp,
                             bank->dt_inv, static_cast<hipStream_t>(stream), sig);
    } else {
        hipLaunchKernelGGL(sg::sg_bank_tick_kernel, dim3(sg::bank_blocks(bank)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           bank->d_ring, d_samples, d_out, bank->streams, bank->d_table, ws, bank->wp, bank->dt_inv, 0, sig);
    }
    if (!sg::hip_ok(hipGetLastError(), who)) return -1;
    bank->wp = (bank->wp + 1) % ws;
    bank->received++;
    if (emit) bank->emitted++;
    return emit;
}

int savgol_streambank_push(SavgolStreamBank *bank, const float *d_samples, float *d_out, void *stream)
{
    return bank_tick(bank, d_samples, d_out, stream, sg::TickSignal{nullptr, nullptr, 0u}, "savgol_streambank_push");
}

// One tick, and its outputs are in d_out (in memory: write-through stores) when the call returns -- without hipStreamSynchronize, which is 5-6 us
// of the 13 us a launch + synchronise tick costs from C.  The tick kernel signals its own completion: the block that finishes last writes a
// sequence number into a word of pinned host memory (tick_signal above) and the host spins on it.  Needs a multiple of 64 streams (whole waves
// at the barrier); otherwise -- and if pinned memory cannot be had -- the call is push + hipStreamSynchronize, same results.
int savgol_streambank_push_wait(SavgolStreamBank *bank, const float *d_samples, float *d_out, void *stream)
{
    const char *who = "savgol_streambank_push_wait";
    if (!bank || !d_samples || !d_out) { sg_set_error("%s: NULL pointer", who); return -1; }
    // arguments and device first, THEN the lazy allocation: a first call from the wrong device used to leave the counter on that device for
    // good, and later calls from the right one then added to a foreign-device pointer
    if (!sg::bank_on_current_device(bank, who)) return -1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (bank->signal_state == 0) {
        void *host = nullptr, *dev = nullptr, *counter = nullptr;
        bank->signal_state = -1;
        if (bank->streams % 64 == 0 && hipHostMalloc(&host, 64, hipHostMallocMapped) == hipSuccess && host) {
            if (hipHostGetDevicePointer(&dev, host, 0) == hipSuccess && dev && hipMalloc(&counter, 64) == hipSuccess && counter &&
                hipMemset(counter, 0, 64) == hipSuccess) {
                bank->signal = static_cast<volatile unsigned *>(host);
                *bank->signal = 0;
                bank->signal_dev = static_cast<unsigned *>(dev);
                bank->signal_counter = static_cast<unsigned *>(counter);
                bank->signal_seq = 0;
                bank->signal_state = 1;
            } else {
                if (counter) (void)hipFree(counter);
                (void)hipHostFree(host);
            }
        }
        (void)hipGetLastError();
    }
    if (bank->signal_state != 1) {
        const int rc = savgol_streambank_push(bank, d_samples, d_out, stream);
        if (rc < 0) return rc;
        return sg::hip_ok(hipStreamSynchronize(st), "savgol_streambank_push_wait: hipStreamSynchronize") ? rc : -1;
    }
    const unsigned seq = ++bank->signal_seq;
    const int rc = bank_tick(bank, d_samples, d_out, stream, sg::TickSignal{bank->signal_counter, bank->signal_dev, seq}, who);
    if (rc < 0) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long spins = 0;
    while (*bank->signal != seq) {
        _mm_pause();
        if ((++spins & 0xfffff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
            // a wedged queue or a faulted kernel: let the synchronise report it, and re-arm the counter for the next tick
            const bool ok = sg::hip_ok(hipStreamSynchronize(st), "savgol_streambank_push_wait: hipStreamSynchronize");
            (void)hipMemset(bank->signal_counter, 0, 64);
            return ok && *bank->signal == seq ? rc : -1;
        }
    }
    // The flag was written after every block's write-through output stores had drained (tick_signal: vmcnt(0) per wave, the block's barrier, the
    // ticket) -- the `sc0 sc1` store + drained-flag hand-off of /opt/skills/guides/MI355X_MICROARCH.md ("Valid forms"), measured on gfx950, not a
    // guarantee of the HIP memory model.  On the host side the acquire below keeps the caller's reads of d_out behind the spin's last load.
    std::atomic_thread_fence(std::memory_order_acquire);
    return rc;
}

int savgol_streambank_push_full(SavgolStreamBank *bank, const float *d_samples, float *d_out, int max_rows, void *stream)
{
    if (!bank || !d_samples || !d_out || max_rows <= 0) { sg_set_error("savgol_streambank_push_full: bad arguments"); return -1; }
    if (!sg::bank_on_current_device(bank, "savgol_streambank_push_full")) return -1;
    const int ws = bank->filter->window_size, n = bank->filter->config.half_window;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool was_filling = bank->received < (unsigned long long)ws;
    const bool fills_now = was_filling && bank->received + 1 >= (unsigned long long)ws;
    if (!fills_now) {
        const int rc = savgol_streambank_push(bank, d_samples, d_out, stream);
        return rc;                                   // 0 while filling, 1 afterwards, -1 on error
    }
    // the tick that completes the window: store the sample, then n leading rows + the centre row
    hipLaunchKernelGGL(sg::sg_bank_tick_kernel, dim3(sg::bank_blocks(bank)), dim3(256), 0, st, bank->d_ring, d_samples,
                       d_out, bank->streams, bank->d_table, ws, bank->wp, bank->dt_inv, 0, sg::TickSignal{nullptr, nullptr, 0u});
    bank->wp = (bank->wp + 1) % ws;
    bank->received++;
    sg::RowList rows; memset(&rows, 0, sizeof(rows));
    for (int e = 0; e < n && rows.count < max_rows; ++e) { rows.row[rows.count] = 1 + e; rows.backward[rows.count] = 1; rows.count++; }
    if (rows.count < max_rows) { rows.row[rows.count] = 0; rows.backward[rows.count] = 0; rows.count++; }
    hipLaunchKernelGGL(sg::sg_bank_rows_kernel, dim3(sg::bank_blocks(bank), rows.count), dim3(256), 0, st, bank->d_ring, d_out,
                       bank->streams, bank->d_table, ws, bank->wp, bank->dt_inv, rows);
    if (!sg::hip_ok(hipGetLastError(), "savgol_streambank_push_full launch")) return -1;
    bank->emitted += rows.count;
    return rows.count;
}

int savgol_streambank_push_block(SavgolStreamBank *bank, const float *d_samples, size_t ticks, float *d_out, void *stream)
{
    if (!bank || !d_samples || !d_out) { sg_set_error("savgol_streambank_push_block: NULL pointer"); return -1; }
    if (!sg::bank_on_current_device(bank, "savgol_streambank_push_block")) return -1;
    if (ticks == 0) return 0;
    const int ws = bank->filter->window_size;
    hipStream_t st = static_cast<hipStream_t>(stream);
    sg::DeviceCtx *ctx = sg::ctx_get();
    if (!ctx) return -1;
    // a launch indexes < 2^31 ticks: longer calls go part by part, the ring kept in step in between (rows of a later part that reach back before it
    // come out of the ring).  Every half window 1..32 has a block kernel (LDS-DMA tiles where the call shape allows, else the walk / register tiles).
    for (size_t done = 0; done < ticks;) {
        const size_t part = ticks - done < ((size_t)1 << 30) ? ticks - done : ((size_t)1 << 30);
        const int wp_part = (int)(((size_t)bank->wp + done) % (size_t)ws);
        if (sg::sg_bank_roll_launch(bank->filter->config.half_window, bank->filter->center_weights, bank->d_ring, d_samples + done * bank->streams,
                                    d_out + done * bank->streams, bank->streams, wp_part, bank->received + done, part, bank->dt_inv,
                                    (bank->flags & SAVGOL_STREAMBANK_FMA) ? 1 : 0, ctx->cu_count, st) != 0) {
            sg_set_error("savgol_streambank_push_block: no kernel for half_window %d", bank->filter->config.half_window);
            return -1;
        }
        // the kernel wrote the outputs; the newest samples still have to reach the ring
        hipLaunchKernelGGL(sg::sg_bank_store_tail_kernel, dim3((unsigned)((bank->streams + 255) / 256), 8), dim3(256), 0, st,
                           bank->d_ring, d_samples + done * bank->streams, bank->streams, ws, wp_part, part);
        done += part;
    }
    if (!sg::hip_ok(hipGetLastError(), "savgol_streambank_push_block launch")) return -1;
    const unsigned long long before = bank->received;
    bank->received += ticks;
    bank->wp = (int)((bank->wp + ticks) % (size_t)ws);
    unsigned long long produced = 0;
    if (bank->received >= (unsigned long long)ws) {
        const unsigned long long first = (before + 1 >= (unsigned long long)ws) ? before + 1 : (unsigned long long)ws;
        produced = bank->received - first + 1;
    }
    bank->emitted += produced;
    return (int)produced;
}

static int bank_edge_rows(SavgolStreamBank *bank, float *d_out, int max_rows, void *stream, bool leading, const char *who)
{
    const int ws = bank->filter->window_size, n = bank->filter->config.half_window;
    if (bank->received < (unsigned long long)ws) return 0;
    sg::RowList rows; memset(&rows, 0, sizeof(rows));
    rows.count = max_rows < n ? max_rows : n;
    for (int i = 0; i < rows.count; ++i) {
        rows.row[i] = leading ? 1 + i : bank->trail_base + (n - 1 - i);
        rows.backward[i] = leading ? 1 : 0;
    }
    hipLaunchKernelGGL(sg::sg_bank_rows_kernel, dim3(sg::bank_blocks(bank), rows.count), dim3(256), 0,
                       static_cast<hipStream_t>(stream), bank->d_ring, d_out, bank->streams, bank->d_table, ws, bank->wp,
                       bank->dt_inv, rows);
    if (!sg::hip_ok(hipGetLastError(), who)) return -1;
    bank->emitted += rows.count;
    return rows.count;
}

int savgol_streambank_flush(SavgolStreamBank *bank, float *d_out, int max_rows, void *stream)
{
    if (!bank || !d_out || max_rows <= 0) { sg_set_error("savgol_streambank_flush: bad arguments"); return -1; }
    if (!sg::bank_on_current_device(bank, "savgol_streambank_flush")) return -1;
    return bank_edge_rows(bank, d_out, max_rows, stream, false, "savgol_streambank_flush launch");
}

int savgol_streambank_flush_leading(SavgolStreamBank *bank, float *d_out, int max_rows, void *stream)
{
    if (!bank || !d_out || max_rows <= 0) { sg_set_error("savgol_streambank_flush_leading: bad arguments"); return 0; }
    if (!sg::bank_on_current_device(bank, "savgol_streambank_flush_leading")) return 0;
    const int rc = bank_edge_rows(bank, d_out, max_rows, stream, true, "savgol_streambank_flush_leading launch");
    return rc < 0 ? 0 : rc;
}

bool   savgol_streambank_ready(const SavgolStreamBank *bank) { return bank && bank->received >= (unsigned long long)bank->filter->window_size; }
size_t savgol_streambank_latency(const SavgolStreamBank *bank) { return bank ? bank->filter->config.half_window : 0; }
size_t savgol_streambank_streams(const SavgolStreamBank *bank) { return bank ? bank->streams : 0; }
size_t savgol_streambank_samples_received(const SavgolStreamBank *bank) { return bank ? (size_t)bank->received : 0; }
size_t savgol_streambank_samples_output(const SavgolStreamBank *bank) { return bank ? (size_t)bank->emitted : 0; }

// checkpoint blob: {wp, received, emitted} header + the ring
struct BankHeader { long long wp; unsigned long long received, emitted; unsigned long long streams; unsigned long long ws; unsigned long long magic; };
static const unsigned long long kBankMagic = 0x5347424b30303032ull;      // "SGBK0002"

size_t savgol_streambank_state_bytes(const SavgolStreamBank *bank)
{
    return bank ? sizeof(BankHeader) + sizeof(float) * (size_t)bank->filter->window_size * bank->streams : 0;
}

int savgol_streambank_save(const SavgolStreamBank *bank, void *host_blob, void *stream)
{
    if (!bank || !host_blob) { sg_set_error("savgol_streambank_save: NULL pointer"); return -1; }
    if (!sg::bank_on_current_device(bank, "savgol_streambank_save")) return -1;
    BankHeader h = {bank->wp, bank->received, bank->emitted, bank->streams, (unsigned long long)bank->filter->window_size, kBankMagic};
    memcpy(host_blob, &h, sizeof(h));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!sg::hip_ok(hipMemcpyAsync(static_cast<char *>(host_blob) + sizeof(h), bank->d_ring,
                                   sizeof(float) * (size_t)h.ws * bank->streams, hipMemcpyDeviceToHost, st), "bank save"))
        return -1;
    return sg::hip_ok(hipStreamSynchronize(st), "bank save") ? 0 : -1;
}

int savgol_streambank_load(SavgolStreamBank *bank, const void *host_blob, void *stream)
{
    if (!bank || !host_blob) { sg_set_error("savgol_streambank_load: NULL pointer"); return -1; }
    if (!sg::bank_on_current_device(bank, "savgol_streambank_load")) return -1;
    BankHeader h;
    memcpy(&h, host_blob, sizeof(h));
    if (h.magic != kBankMagic) { sg_set_error("savgol_streambank_load: not a stream-bank blob (bad magic)"); return -1; }
    if (h.streams != bank->streams || h.ws != (unsigned long long)bank->filter->window_size) {
        sg_set_error("savgol_streambank_load: blob is for %llu streams / window %llu", h.streams, h.ws);
        return -1;
    }
    // the write position indexes the ring in every later kernel: it must be inside it and consistent with the counters
    if (h.wp < 0 || (unsigned long long)h.wp >= h.ws || (unsigned long long)h.wp != h.received % h.ws || h.emitted > h.received) {
        sg_set_error("savgol_streambank_load: corrupt blob (write position %lld, %llu received, %llu emitted, window %llu)", h.wp, h.received,
                     h.emitted, h.ws);
        return -1;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!sg::hip_ok(hipMemcpyAsync(bank->d_ring, static_cast<const char *>(host_blob) + sizeof(h),
                                   sizeof(float) * (size_t)h.ws * bank->streams, hipMemcpyHostToDevice, st), "bank load"))
        return -1;
    if (!sg::hip_ok(hipStreamSynchronize(st), "bank load")) return -1;
    bank->wp = (int)h.wp; bank->received = h.received; bank->emitted = h.emitted;
    return 0;
}

}  // extern "C"
