// sg_api_1d.cpp -- host side of the 1-D path: the five drop-in entry points of savgolFilter.h and
// the device-pointer batch entry points of savgol_hip.h.  All arithmetic on samples happens in the
// HIP kernels (sg_k1d.hpp); this file validates, picks tile geometry and enqueues.
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "sg_k1d_host.hpp"
#include "sg_runtime.hpp"

using sg::DeviceCtx;

// ------------------------------------------------------------------------------------------------
// lifecycle (host only) -- reference savgol_create / savgol_destroy, src/savgolFilter.c:688-728
// ------------------------------------------------------------------------------------------------
extern "C" SavgolFilter *savgol_create(const SavgolConfig *config)
{
    if (!config) return nullptr;
    const int n = config->half_window, m = config->poly_order, d = config->derivative;
    // same order of checks and the same messages as validate_config (:639-677)
    if (n == 0 || n > SAVGOL_MAX_HALF_WINDOW) {
        fprintf(stderr, "savgol: half_window must be in [1, %d], got %d\n", SAVGOL_MAX_HALF_WINDOW, n);
        return nullptr;
    }
    const int ws = 2 * n + 1;
    if (m >= ws) {
        fprintf(stderr, "savgol: poly_order must be < window_size (%d), got %d\n", ws, m);
        return nullptr;
    }
    if (d > SAVGOL_MAX_DERIVATIVE) {
        fprintf(stderr, "savgol: derivative must be ≤ %d, got %d\n", SAVGOL_MAX_DERIVATIVE, d);
        return nullptr;
    }
    if (d > m) {
        fprintf(stderr, "savgol: derivative (%d) cannot exceed poly_order (%d)\n", d, m);
        return nullptr;
    }
    if (!(config->time_step > 0.0f)) {
        fprintf(stderr, "savgol: time_step must be > 0, got %f\n", config->time_step);
        return nullptr;
    }
    if (!sg_weights_valid(n, m, d, config->time_step)) {
        fprintf(stderr, "savgol: half_window %d with poly_order %d exceeds the factorial table (2n+m+1 must be < 76)\n", n, m);
        return nullptr;
    }
    SavgolFilter *f = static_cast<SavgolFilter *>(calloc(1, sizeof(SavgolFilter)));
    if (!f) {
        fprintf(stderr, "savgol: failed to allocate filter context\n");
        return nullptr;
    }
    f->config = *config;
    sg_weights_fill(f);
    return f;
}

extern "C" void savgol_destroy(SavgolFilter *filter) { free(filter); }

// ------------------------------------------------------------------------------------------------
// enqueue one batch
// ------------------------------------------------------------------------------------------------
namespace {

// Process-wide DEFAULTS of the per-call flags (savgol_hip_set_option); the *_ex entry points take the flags themselves, so two
// threads can run different summation orders / tile widths at the same time.
std::atomic<unsigned> g_default_flags{0};

// Everything a batch call needs from a filter besides the samples, built once per distinct filter CONTENT and device: the
// uploaded edge rows / reference-order table / moment table, the polynomial fit behind the moment table, the by-value taps and
// the fp64 symmetry check.  Rounds 1-2 re-packed the edge rows (8.3 KB at n = 32), hashed them byte-wise and scanned the table
// cache linearly on EVERY call, then scanned the moment fits under a second mutex: ~10 us per call at n = 32 (tools/time_batch_host.c).
// A SavgolFilter is a caller-owned POD (callers do edit tables by hand), so the key is the content -- a word-wise hash of the
// bytes the kernels read, ~0.2 us -- and a hit compares them.  Entries live as long as the process (queued launches and captured
// graphs keep the device pointers).
struct FilterPlan {
    int    n = 0, ws = 0, device = -1;
    std::vector<unsigned char> content;             // SavgolFilter bytes [0, used_bytes)
    sg::Taps taps32, taps64;
    int    sym = 0;                                 // fp64 path: 1 = centre taps (anti)symmetric bit for bit, 0 = tap sym_bad is not
    int    sym_bad = 0;
    bool   odd = false;
    const float *d_edges = nullptr, *d_ref = nullptr, *d_moment = nullptr;
    int    moment_terms = -1;                       // -1: not fitted yet
    float  moment_table[sg::MOMENT_TABLE_FLOATS];
    const double *d_moment64 = nullptr;             // the opt-in fp64 block-moment path (SAVGOL_BATCH_MOMENT_F64)
    int    moment64_terms = -1;
    double moment64_table[sg::MOMENT64_TABLE_DOUBLES];
};
std::mutex g_plan_mu;
std::unordered_multimap<uint64_t, FilterPlan *> g_plans;

inline size_t filter_used_bytes(const SavgolFilter *f)
{
    return offsetof(SavgolFilter, edge_weights) + sizeof(f->edge_weights[0]) * (size_t)f->config.half_window;
}

// the plan of (filter content, device); `need` = which lazily built parts this call wants
enum : unsigned { NEED_EDGES = 1, NEED_REF = 2, NEED_MOMENT = 4, NEED_MOMENT64 = 8 };
const FilterPlan *plan_get(DeviceCtx *ctx, const SavgolFilter *f, unsigned need)
{
    const size_t bytes = filter_used_bytes(f);
    const uint64_t key = sg::hash64(f, bytes, 0x1d00u + (uint64_t)ctx->ordinal);
    std::lock_guard<std::mutex> lock(g_plan_mu);
    FilterPlan *p = nullptr;
    auto range = g_plans.equal_range(key);
    for (auto it = range.first; it != range.second; ++it)
        if (it->second->device == ctx->ordinal && it->second->content.size() == bytes && memcmp(it->second->content.data(), f, bytes) == 0) { p = it->second; break; }
    const int n = f->config.half_window, ws = f->window_size;
    if (!p) {
        p = new FilterPlan();
        p->n = n; p->ws = ws; p->device = ctx->ordinal;
        p->content.assign(reinterpret_cast<const unsigned char *>(f), reinterpret_cast<const unsigned char *>(f) + bytes);
        memset(&p->taps32, 0, sizeof(p->taps32));
        memset(&p->taps64, 0, sizeof(p->taps64));
        memcpy(p->taps32.w, f->center_weights, sizeof(float) * ws);
        // The fp64 kernel keeps taps 0..n as doubles in SGPRs and takes tap 2n-k = +-tap k from them.  Tables built by
        // savgol_create are (anti)symmetric bit for bit (only Gram terms of the derivative's parity are non-zero at
        // t = 0); a hand-edited table that is not cannot run on this path.
        p->odd = (f->config.derivative & 1) != 0;
        p->sym = 1;
        for (int k = 0; k <= n; ++k) {
            const float a = f->center_weights[k], b = f->center_weights[2 * n - k];
            if (!(p->odd ? (a == -b) : (a == b))) { p->sym = 0; p->sym_bad = k; break; }
            p->taps64.wd[k] = (double)a;
        }
        g_plans.emplace(key, p);
    }
    if ((need & NEED_EDGES) && !p->d_edges) {
        float packed[SAVGOL_MAX_HALF_WINDOW * SAVGOL_MAX_WINDOW];               // rows packed [n][ws]
        for (int e = 0; e < n; ++e) memcpy(packed + e * ws, f->edge_weights[e], sizeof(float) * ws);
        p->d_edges = sg::ctx_table(ctx, packed, sizeof(float) * n * ws, 0x1d00u + (unsigned)n);
        if (!p->d_edges) return nullptr;
    }
    if ((need & NEED_REF) && !p->d_ref) {
        float packed[(SAVGOL_MAX_HALF_WINDOW + 1) * SAVGOL_MAX_WINDOW];         // centre row, then the edge rows
        memcpy(packed, f->center_weights, sizeof(float) * ws);
        for (int e = 0; e < n; ++e) memcpy(packed + (size_t)(1 + e) * ws, f->edge_weights[e], sizeof(float) * ws);
        p->d_ref = sg::ctx_table(ctx, packed, sizeof(float) * (size_t)(n + 1) * ws, 0x1e00u + (unsigned)n);
        if (!p->d_ref) return nullptr;
    }
    if ((need & NEED_MOMENT) && p->moment_terms < 0) {
        // wide-window fast path (half windows 24..32): the polynomial fit of the centre taps (sg_k1d_moment_fit.cpp)
        const int terms = sg1d_momenth_prepare(n, f->center_weights, p->moment_table);            // the half-lane form (sg_k1d_momenth.hpp)
        if (terms > 0) {
            p->d_moment = sg::ctx_table(ctx, p->moment_table, sizeof(p->moment_table), 0x2100u + (unsigned)n);
            if (!p->d_moment) return nullptr;
        }
        p->moment_terms = terms;
    }
    if ((need & NEED_MOMENT64) && p->moment64_terms < 0) {
        const int terms = sg1d_moment64_prepare(n, f->center_weights, p->moment64_table);
        if (terms > 0) {
            p->d_moment64 = reinterpret_cast<const double *>(sg::ctx_table(ctx, p->moment64_table, sizeof(p->moment64_table), 0x2000u + (unsigned)n));
            if (!p->d_moment64) return nullptr;
        }
        p->moment64_terms = terms;
    }
    return p;
}

enum Variant { FULL = 0, VALID = 1, FULL_POLY_EDGES = 2 /* strided: polynomial edges whatever the mode */,
               INTERIOR = 3 /* out[g] for g in [n, L-n) only: the segments of a channel longer than one launch indexes */ };

inline float dt_inverse(const SavgolFilter *f)      // reference :759
{
    return (f->dt_scale != 0.0f) ? (1.0f / f->dt_scale) : 1.0f;
}

bool filter_sane(const SavgolFilter *f, const char *who)
{
    const int n = f->config.half_window;
    if (n < 1 || n > SAVGOL_MAX_HALF_WINDOW || f->window_size != 2 * n + 1) {
        sg_set_error("%s: filter struct is not a valid SavgolFilter (half_window %d, window_size %d)", who, n,
                     f->window_size);
        return false;
    }
    return true;
}

// samples per channel one launch indexes (32-bit sample and tile indices inside the kernels); longer channels are cut into
// sub-rows by enqueue_long below -- the reference's savgol_apply takes a size_t length
constexpr size_t LAUNCH_MAX_LENGTH = (size_t)1 << 30;

template <typename T>
int enqueue_long(const char *who, const SavgolFilter *f, const T *d_in, T *d_out, size_t channels, size_t length,
                 size_t in_ld, size_t out_ld, Variant variant, hipStream_t st, unsigned flags);

// d_in / d_out as pitched row sets [base + c ld, base + c ld + len): do any two rows share a byte?  The extents may interleave
// (in = buf[:, 0, :], out = buf[:, 1, :] with both pitches 2 L) without any row touching another.
template <typename T>
bool rows_overlap(const T *d_in, size_t in_ld, size_t in_len, const T *d_out, size_t out_ld, size_t out_len, size_t channels)
{
    const uintptr_t a0 = (uintptr_t)d_in, a1 = a0 + ((channels - 1) * in_ld + in_len) * sizeof(T);
    const uintptr_t b0 = (uintptr_t)d_out, b1 = b0 + ((channels - 1) * out_ld + out_len) * sizeof(T);
    if (!(a0 < b1 && b0 < a1)) return false;                       // the whole extents are disjoint: the common case
    if (in_ld != out_ld || channels == 1) return true;
    // equal pitches: row i of one buffer can only meet rows of the other that start within one pitch of it
    const uintptr_t pitch = in_ld * sizeof(T);
    const uintptr_t d = a0 <= b0 ? (b0 - a0) % pitch : (pitch - (a0 - b0) % pitch) % pitch;      // offset of an output row inside the input's pitch
    // input rows occupy [0, in_len) mod pitch, output rows [d, d + out_len) mod pitch
    const uintptr_t il = in_len * sizeof(T), ol = out_len * sizeof(T);
    return d < il || d + ol > pitch;
}

// a stream-ordered scratch block that every exit hands back (the error returns of the in-place call leaked its stash)
template <typename T>
struct ScratchGuard {
    T *p; hipStream_t st; bool armed;
    ~ScratchGuard() { if (armed && p) (void)sg::scratch_free(p, st, "scratch free (in-place stash, error exit)"); }
};

template <typename T>
int enqueue_batch(const char *who, const SavgolFilter *f, const T *d_in, T *d_out, size_t channels, size_t length,
                  size_t in_ld, size_t out_ld, Variant variant, hipStream_t st, unsigned flags)
{
    const bool reference_order = sizeof(T) == 4 && (flags & SAVGOL_BATCH_REFERENCE_SUMMATION) != 0;
    const bool correct_edge = (flags & SAVGOL_BATCH_CORRECT_LEADING_EDGE) != 0;
    if (!f || !d_in || !d_out) { sg_set_error("%s: NULL pointer", who); return -1; }
    if (!filter_sane(f, who)) return -1;
    const int n = f->config.half_window, ws = f->window_size;
    if (length < (size_t)ws) { sg_set_error("%s: data length (%zu) < window size (%d)", who, length, ws); return -1; }
    if (length > LAUNCH_MAX_LENGTH) return enqueue_long<T>(who, f, d_in, d_out, channels, length, in_ld, out_ld, variant, st, flags);
    const size_t out_len = (variant == VALID) ? length - 2 * (size_t)n : length;
    if (in_ld < length || out_ld < out_len) { sg_set_error("%s: row pitch smaller than the row", who); return -1; }
    if (channels == 0) return 0;
    // IN PLACE (round 5; the reference advertises output == input, include/iterative/savgolFilter.h:148): exactly the same
    // rows -- same base, same pitch, the full-length variants -- run on the tile kernels with every tile's halo taken from a stash that is
    // filled first (3 % of the data at n = 32), and give the OUT-OF-PLACE answer, not the reference's own in-place result (which reads samples it
    // has already overwritten: documented divergence, as for the host-pointer call).  Any other overlap still races and is refused.
    const bool inplace = static_cast<const void *>(d_in) == static_cast<const void *>(d_out) && in_ld == out_ld && (variant == FULL || variant == FULL_POLY_EDGES);
    if (!inplace && rows_overlap(d_in, in_ld, length, d_out, out_ld, out_len, channels)) {
        sg_set_error("%s: d_in and d_out overlap (in place means the SAME rows: d_out == d_in with the same pitch; partial overlaps race)", who);
        return -1;
    }

    DeviceCtx *ctx = sg::ctx_get();
    if (!ctx) return -1;
    const int mode = (variant == FULL) ? (int)f->config.boundary : (int)SAVGOL_BOUNDARY_POLYNOMIAL;
    const bool poly = (mode == SAVGOL_BOUNDARY_POLYNOMIAL);
    const bool want_edges = poly && variant != VALID && variant != INTERIOR;
    // Block moments only where they cost no accuracy (round 4, tools/diag_1d_accuracy.py): the block's sum is the largest single term of
    // an output, so where the filter nulls the signal -- moving averages (poly_order 0, 1) on a tone, second derivatives -- its rounding
    // shows: 1.5-2.9 x the reference's own fp32 error there, <= 1.3 x for poly_order >= 2 with derivative <= 1 (every half window
    // 24..32, all boundary modes; the plain three-chain kernel: 0.7-1.3 x everywhere).  Those filters take the plain kernel (8 % slower).
    const bool moment_safe = f->config.poly_order >= 2 && f->config.derivative <= 1;
    const bool want_moment = sizeof(T) == 4 && !reference_order && n >= sg::MOMENTH_MIN_N && n <= sg::MOMENT_MAX_N &&
                             !(flags & SAVGOL_BATCH_PLAIN_SUMMATION) && moment_safe;
    // fp64, half windows 24..32, on request only: block moments (sg_k1d_moment64.hpp) -- within ~1e-7 of the default path, not its 1e-12
    const bool want_moment64 = sizeof(T) == 8 && (flags & SAVGOL_BATCH_MOMENT_F64) && n >= sg::MOMENT_MIN_N && n <= sg::MOMENT_MAX_N;
    if (inplace && reference_order) {
        // the reference-order kernels read their whole window from the rows: run them from a stream-ordered copy of the input
        T *copy = static_cast<T *>(sg::scratch_alloc(ctx, channels * in_ld * sizeof(T), st, "scratch (in-place copy for the reference-order kernels)"));
        if (!copy) return -1;
        int rc = sg::hip_ok(hipMemcpyAsync(copy, d_in, ((channels - 1) * in_ld + length) * sizeof(T), hipMemcpyDeviceToDevice, st), "in-place copy") ? 0 : -1;
        if (rc == 0) rc = enqueue_batch<T>(who, f, copy, d_out, channels, length, in_ld, out_ld, variant, st, flags);
        if (!sg::scratch_free(copy, st, "scratch free (in-place copy)")) rc = -1;
        return rc;
    }
    const FilterPlan *plan = plan_get(ctx, f, reference_order ? NEED_REF : ((want_edges ? NEED_EDGES : 0u) | (want_moment ? NEED_MOMENT : 0u) | (want_moment64 ? NEED_MOMENT64 : 0u)));
    if (!plan) return -1;

    if constexpr (sizeof(T) == 4) {
        if (reference_order) {
            // one output per thread in the reference's own summation order: bit-identical to its savgol_apply
            const float *d_table = plan->d_ref;
            const int rmode = mode;
            const bool inner = variant == VALID || variant == INTERIOR;                 // no edge outputs at all
            const int lo = inner ? n : 0, hi = inner ? (int)length - n : (int)length;
            const int shift = (variant == VALID) ? n : 0;
            const int negate = (poly && correct_edge && (f->config.derivative & 1)) ? 1 : 0;
            int rc;
            if (channels * length >= ((size_t)1 << 16) && length >= (size_t)4 * ws) {
                // long batches: the packed kernel for everything the centre taps produce, the per-thread kernel for
                // the 2n POLYNOMIAL edge samples of every channel
                const int clo = (poly || inner) ? n : 0, chi = (poly || inner) ? (int)length - n : (int)length;
                rc = sg1d_launch_refpk_f32(d_in, d_out, (long long)in_ld, (long long)out_ld, (long long)length, n, f->center_weights,
                                           dt_inverse(f), rmode, clo, chi, shift, channels, ctx->cu_count, st);
                if (rc == 0 && poly && !inner) {
                    rc = sg1d_launch_reference_order_f32(d_in, d_out, (long long)in_ld, (long long)out_ld, (long long)length, n, d_table,
                                                         dt_inverse(f), rmode, 0, n, 0, negate, channels, st);
                    if (rc == 0)
                        rc = sg1d_launch_reference_order_f32(d_in, d_out, (long long)in_ld, (long long)out_ld, (long long)length, n, d_table,
                                                             dt_inverse(f), rmode, (int)length - n, (int)length, 0, 0, channels, st);
                }
            } else {
                rc = sg1d_launch_reference_order_f32(d_in, d_out, (long long)in_ld, (long long)out_ld, (long long)length, n, d_table,
                                                     dt_inverse(f), rmode, lo, hi, shift, negate, channels, st);
            }
            if (rc != 0) { sg_set_error("%s: kernel launch failed", who); return -1; }
            return 0;
        }
    }

    constexpr int E = 16 / (int)sizeof(T);
    // tile width: the wide tile where one is built and the batch is big enough to keep the chip in whole rounds of them
    int vpl = sg::vectors_per_lane(sizeof(T), f->config.half_window);
    const int vpl_wide = sg::wide_vectors_per_lane(sizeof(T), f->config.half_window);
    const int tile_mode = (flags & SAVGOL_BATCH_TILE_NARROW) ? 1 : ((flags & SAVGOL_BATCH_TILE_WIDE) ? 2 : 0);
    const int wide = vpl_wide != vpl && tile_mode != 1 && !(want_moment64 && plan->moment64_terms > 0 && plan->sym) &&       // (the moment kernels are laid out for 8 vectors per lane)
                     !(want_moment && plan->moment_terms > 0) &&
                     (tile_mode == 2 || (unsigned long long)channels * ((length + 64u * vpl_wide * E - 1) / (64u * vpl_wide * E)) >= sg::WIDE_TILE_MIN_TILES);
    if (wide) vpl = vpl_wide;
    const unsigned TW = 64u * (unsigned)vpl * E;

    sg::Job1D job;
    memset(&job, 0, sizeof(job));
    // Tile order: chunks of 64 blocks dealt to the XCDs round robin (round 5; until then every XCD swept one contiguous eighth of the tiles, its
    // front gigabytes away from the other seven).  Sampled over fresh placements of the two buffers inside one process, columns rotated
    // (tools/placement_1d.py, profiles/r05_placement_1d.txt): headline shape 5.53 against 5.65 ms median, ahead on every one of six buffer pairs; chunks
    // of 32 / 256 / 1024 blocks 5.64 / 5.60 / 5.60; launch order loses 15 %.  SAVGOL_HIP_1D_XCD_CHUNK_LOG2=0 is the old order.
    static const unsigned xcd_chunk_env = [] { const char *e = getenv("SAVGOL_HIP_1D_XCD_CHUNK_LOG2"); return e ? (unsigned)atoi(e) : 6u; }();
    job.xcd_chunk_log2 = xcd_chunk_env;
    job.in_ld = (long long)in_ld;
    job.out_ld = (long long)out_ld;
    job.length = (unsigned)length;
    sg::set_tiles_per_channel(job, (unsigned)((length + TW - 1) / TW));
    job.dt_inv = dt_inverse(f);
    // which samples the centre kernel stores: the interior when edge rows / VALID take the rest
    const bool interior_only = (variant == VALID) || poly;              // INTERIOR runs as poly without the edge rows
    job.store_lo = interior_only ? (unsigned)n : 0u;
    job.store_hi = interior_only ? (unsigned)(length - n) : (unsigned)length;
    job.out_shift = (variant == VALID) ? (unsigned)n : 0u;
    job.flags = ((unsigned)mode & sg::JOB_MODE_MASK);
    if (mode < 0 || mode > 255) job.flags = 255u;                       // unknown mode: zero padding (reference :478-480)
    if (job.dt_inv != 1.0f) job.flags |= sg::JOB_SCALE;
    if (((uintptr_t)d_in % 16 == 0) && (in_ld % E == 0)) job.flags |= sg::JOB_VEC_IN;
    if (((uintptr_t)d_out % 16 == 0) && (out_ld % E == 0) && (job.out_shift % E == 0)) job.flags |= sg::JOB_VEC_OUT;
    if (sizeof(T) == 4 && f->config.derivative >= 1) {
        // fp32 derivative filters run on centred tiles (sg1d_tile_body, JOB_CENTRE): what a constant comes out as is taken from the reference's own table
        double wsum = 0.0;
        for (int k = 0; k <= 2 * n; ++k) wsum += (double)f->center_weights[k];
        job.centre_sum = (float)wsum;
        job.flags |= sg::JOB_CENTRE;
    }

    if (sizeof(T) == 8) {
        if (!plan->sym) {
            sg_set_error("%s: fp64 path needs the centre taps savgol_create builds (tap[k] == %stap[2n-k]); tap %d is not", who,
                         plan->odd ? "-" : "", plan->sym_bad);
            return -1;
        }
        if (plan->odd) job.flags |= sg::JOB_ODD_TAPS;
    }
    const sg::Taps &taps = sizeof(T) == 8 ? plan->taps64 : plan->taps32;
    const float *d_edges = want_edges ? plan->d_edges : nullptr;
    // half windows 24..32, fp32: block moments replace the taps on the lanes' common block when the table is a polynomial
    const float *d_moment = (want_moment && plan->moment_terms > 0) ? plan->d_moment : nullptr;
    const int moment_terms = d_moment ? plan->moment_terms : 0;
    const double *d_moment64 = (want_moment64 && plan->moment64_terms > 0 && plan->sym) ? plan->d_moment64 : nullptr;     // a table that is not a polynomial keeps the plain kernel

    // the POLYNOMIAL edge rows ride in the same launch: two more items per channel behind the tiles (sg1d_edge_item)
    if (d_edges) {
        job.edges = d_edges;
        if (correct_edge && (f->config.derivative & 1)) job.flags |= sg::JOB_EDGE_NEGATE;
    }
    // split so that a launch stays below 2^24 blocks of four tiles (sg::MAX_TILES_PER_LAUNCH)
    size_t max_ch = (size_t)sg::MAX_TILES_PER_LAUNCH / ((size_t)job.tiles_per_channel + 2);
    if (inplace) {
        // in place: channel groups whose stash (one slot of 2 NA samples per ODD tile + 4 NA per channel) stays below 3/4 of what the scratch pool keeps
        // (192 of 256 MiB by default; SAVGOL_HIP_SCRATCH_KEEP_MB moves both), so it is neither handed back to the driver nor mapped again from call to call
        // (round 5: a 1 GiB stash per 1024-channel chunk cost 14 ms a call on most boxes and 98 on one).  The groups follow each other on the stream; config 5's
        // chunk is 6 groups (round 6's first version stashed a slot per tile under a 64 MiB cap: 32 groups of 0.19 ms launches, +22 %).
        const int NA0 = (n + E - 1) / E * E;
        const size_t per_ch = ((size_t)(job.tiles_per_channel / 2) * (size_t)(2 * NA0) + (size_t)(4 * NA0)) * sizeof(T);
        size_t cap = (size_t)(sg::scratch_keep_bytes() / 4 * 3);
        if (cap < ((size_t)64 << 20)) cap = (size_t)64 << 20;
        const size_t group = per_ch ? cap / per_ch : max_ch;
        if (group >= 1 && group < max_ch) max_ch = group;
    }
    // the tile launch of the chosen kernel family: one tile per wave, blocks dispatched in order (see sg1d_center_kernel)
    auto launch_tiles = [&](const sg::Job1D &j) -> int {
        unsigned blocks = (j.total_tiles + j.edge_items + 3u) / 4u;
        blocks = (blocks + 7u) & ~7u;                                    // the XCD remap wants a multiple of 8
        if (d_moment)
            return moment_terms == 3 ? sg1d_launch_f32_momenth_t3(n, &j, d_moment, blocks, st)
                 : moment_terms == 5 ? sg1d_launch_f32_momenth_t5(n, &j, d_moment, blocks, st)
                                     : sg1d_launch_f32_momenth_t7(n, &j, d_moment, blocks, st);
        if (d_moment64) {
            const int mt = plan->moment64_terms;
            return mt == 3 ? sg1d_launch_f64_moment_t3(n, &j, d_moment64, blocks, st)
                 : mt == 5 ? sg1d_launch_f64_moment_t5(n, &j, d_moment64, blocks, st)
                           : sg1d_launch_f64_moment_t7(n, &j, d_moment64, blocks, st);
        }
        return sg::launch_center<T>(n, wide, j, taps, blocks, st);
    };
    const unsigned tpc = job.tiles_per_channel;
    // in place: the 2n+1 input samples of every channel end, for the edge rows that run after the last group (2 (2n+1) samples per channel)
    T *edge_all = nullptr;
    if (inplace && d_edges) {
        edge_all = static_cast<T *>(sg::scratch_alloc(ctx, (channels * 2 * (size_t)ws + 4) * sizeof(T), st, "scratch (in-place edge samples)"));
        if (!edge_all) return -1;
    }
    ScratchGuard<T> edge_guard{edge_all, st, edge_all != nullptr};
    for (size_t c0 = 0; c0 < channels; c0 += max_ch) {
        const size_t nc = (channels - c0 < max_ch) ? channels - c0 : max_ch;
        job.in = d_in + c0 * in_ld;
        job.out = d_out + c0 * out_ld;
        if (!inplace) {
            job.total_tiles = (unsigned)(nc * tpc);
            job.edge_items = d_edges ? (unsigned)(2 * nc) : 0u;
            if (launch_tiles(job) != 0) return -1;
            continue;
        }
        // IN PLACE, two colour phases (round 6; round 5 copied every tile's halo in a pass of its own first).
        // 1. sg1d_launch_ends: the halos that reach past a channel's end + the edge rows' samples -- a few hundred samples per channel;
        // 2. the EVEN tiles: halos live from the rows (their neighbours are untouched), their own first / last NA inputs into the odd neighbours' slots;
        // 3. the ODD tiles: body live, halos from their slots;  4. after the last group: the POLYNOMIAL edge rows of every channel in ONE launch (they
        //    overwrite samples the end tiles read; their 2n+1 input samples per end were put aside by step 1).
        const int NA = (n + E - 1) / E * E;
        const size_t halo = nc * (size_t)(tpc / 2) * (size_t)(2 * NA), ends = nc * (size_t)(4 * NA);
        T *stash = static_cast<T *>(sg::scratch_alloc(ctx, (halo + ends + 4) * sizeof(T), st, "scratch (in-place halo stash)"));
        if (!stash) return -1;
        ScratchGuard<T> guard{stash, st, true};
        if (sg1d_launch_ends(job.in, job.in_ld, job.length, tpc, (int)TW, NA, (int)(job.flags & sg::JOB_MODE_MASK), stash, stash + halo,
                             edge_all ? edge_all + c0 * 2 * (size_t)ws : nullptr, ws, nc, (int)sizeof(T), st) != 0) { sg_set_error("%s: channel-end stash launch failed", who); return -1; }
        sg::Job1D pj = job;
        pj.stash = stash;
        pj.ends = stash + halo;
        pj.edge_stash = nullptr;
        pj.edge_items = 0;
        pj.tpc_all = tpc;
        for (unsigned phase = 1; phase <= 2; ++phase) {
            const unsigned per = phase == 1 ? (tpc + 1u) / 2u : tpc / 2u;
            if (per == 0) continue;
            pj.phase = phase;
            sg::set_tiles_per_channel(pj, per);
            pj.total_tiles = (unsigned)(nc * per);
            if (launch_tiles(pj) != 0) return -1;
        }
        guard.armed = false;
        if (!sg::scratch_free(stash, st, "scratch free (in-place halo stash)")) return -1;
    }
    if (inplace && edge_all) {
        // every channel's two edge items, 2^22 items (two million channels) per launch
        for (size_t c0 = 0; c0 < channels; c0 += (size_t)1 << 21) {
            const size_t nc = channels - c0 < ((size_t)1 << 21) ? channels - c0 : ((size_t)1 << 21);
            sg::Job1D ej = job;
            ej.in = d_in + c0 * in_ld;
            ej.out = d_out + c0 * out_ld;
            ej.edge_stash = edge_all + c0 * 2 * (size_t)ws;
            ej.total_tiles = 0; ej.edge_items = (unsigned)(2 * nc);
            unsigned eb = (ej.edge_items + 3u) / 4u;
            eb = (eb + 7u) & ~7u;
            if (sg::launch_center<T>(n, 0, ej, taps, eb, st) != 0) return -1;
        }
        edge_guard.armed = false;
        if (!sg::scratch_free(edge_all, st, "scratch free (in-place edge samples)")) return -1;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Channels longer than LAUNCH_MAX_LENGTH samples (the reference's savgol_apply takes any size_t length, savgolFilter.c:743-766).
// Outputs [n, L-n) do not depend on the boundary mode: they are enqueued as sub-rows [a-n, b+n) of at most 2^29 + 2n samples
// through the same kernels (what lies beyond a sub-row only reaches outputs it does not store).  The 2n edge outputs of a
// channel only see the 3n samples next to them -- or, PERIODIC, the other end of the channel -- so they are computed on two
// 256-sample copies of the channel ends (one ring of both ends for PERIODIC) and copied into place: same arithmetic per
// output as the unsplit call, in every summation mode.  Everything is enqueued on `st`; the scratch is stream-ordered.
// ------------------------------------------------------------------------------------------------
template <typename T>
int enqueue_long(const char *who, const SavgolFilter *f, const T *d_in, T *d_out, size_t channels, size_t length,
                 size_t in_ld, size_t out_ld, Variant variant, hipStream_t st, unsigned flags)
{
    const size_t n = (size_t)f->config.half_window;
    const size_t out_len = (variant == VALID) ? length - 2 * n : length;
    if (in_ld < length || out_ld < out_len) { sg_set_error("%s: row pitch smaller than the row", who); return -1; }
    if (channels == 0) return 0;
    if (rows_overlap(d_in, in_ld, length, d_out, out_ld, out_len, channels)) { sg_set_error("%s: d_in and d_out overlap (channels longer than 2^30 samples run out of place only: include/savgol_hip.h, in-place contract)", who); return -1; }
    constexpr size_t SEG = (size_t)1 << 29;                       // outputs per sub-row (a multiple of the vector width: sub-rows stay 16-byte aligned)
    for (size_t a = n; a < length - n; a += SEG) {
        const size_t b = (length - n - a < SEG) ? length - n : a + SEG;
        // VALID stores out[g - n], INTERIOR out[g]: the same pointer offset serves both
        if (enqueue_batch<T>(who, f, d_in + (a - n), d_out + (a - n), channels, (b - a) + 2 * n, in_ld, out_ld,
                             variant == VALID ? VALID : INTERIOR, st, flags) != 0) return -1;
    }
    if (variant == VALID || variant == INTERIOR) return 0;

    constexpr size_t E = 256;                                    // samples of a channel end that are copied out (>= 4 half windows, vector aligned)
    const int mode = (variant == FULL) ? (int)f->config.boundary : (int)SAVGOL_BOUNDARY_POLYNOMIAL;
    T *scratch = nullptr;
    DeviceCtx *ctx_long = sg::ctx_get();
    if (!ctx_long) return -1;
    scratch = static_cast<T *>(sg::scratch_alloc(ctx_long, 4 * channels * E * sizeof(T), st, "scratch (channel ends)"));
    if (!scratch) return -1;
    T *in_a = scratch, *in_b = scratch + channels * E, *out_a = scratch + 2 * channels * E, *out_b = scratch + 3 * channels * E;
    auto copy2d = [&](T *dst, size_t dst_ld, const T *src, size_t src_ld, size_t width) {
        return sg::hip_ok(hipMemcpy2DAsync(dst, dst_ld * sizeof(T), src, src_ld * sizeof(T), width * sizeof(T), channels, hipMemcpyDeviceToDevice, st),
                          "hipMemcpy2DAsync(channel ends)");
    };
    bool ok;
    if (mode == (int)SAVGOL_BOUNDARY_PERIODIC) {
        // ring of both ends: [in[L-128 .. L), in[0 .. 128)]; its interior outputs 128-n .. 128+n-1 are the channel's L-n .. L-1, 0 .. n-1
        ok = copy2d(in_a, E, d_in + (length - E / 2), in_ld, E / 2) && copy2d(in_a + E / 2, E, d_in, in_ld, E / 2) &&
             enqueue_batch<T>(who, f, in_a, out_a, channels, E, E, E, INTERIOR, st, flags) == 0 &&
             copy2d(d_out + (length - n), out_ld, out_a + (E / 2 - n), E, n) && copy2d(d_out, out_ld, out_a + E / 2, E, n);
    } else {
        ok = copy2d(in_a, E, d_in, in_ld, E) && copy2d(in_b, E, d_in + (length - E), in_ld, E) &&
             enqueue_batch<T>(who, f, in_a, out_a, 2 * channels, E, E, E, variant, st, flags) == 0 &&      // in_b / out_b follow in_a / out_a
             copy2d(d_out, out_ld, out_a, E, n) && copy2d(d_out + (length - n), out_ld, out_b + (E - n), E, n);
    }
    if (!sg::scratch_free(scratch, st, "scratch free (channel ends)")) ok = false;
    return ok ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------
// Long host-pointer signals: upload, filter and download in chunks so that the two directions of the host link run at the
// same time (round 1 ran H2D -> kernel -> D2H back to back: 6.9 Gsamples/s at 2^26 samples, half of a full-duplex link).
//   main thread : for every chunk  H2D(samples up to the chunk's right halo) -> reference-order kernel on [a_k, b_k) -> event
//   helper      : for every chunk  wait(event) -> D2H of [a_k, b_k) into the caller's buffer
// then the first / last 32 outputs (every boundary mode's edge handling) by the per-thread reference-order kernel.  Chunks
// are sub-rows handed to the same kernels the unchunked call uses (pointer + 32-sample halo; what lies beyond a sub-row only
// reaches outputs that are not stored), so the result is the reference's, bit for bit, as before.
// ------------------------------------------------------------------------------------------------
constexpr size_t PIPE_MIN_LENGTH = (size_t)1 << 23;       // below this the plain path is as fast
constexpr size_t PIPE_CHUNK_DEFAULT = (size_t)1 << 22;    // 16 MB of samples per chunk (SAVGOL_HIP_PIPE_CHUNK_LOG2 overrides: tuning)
constexpr int    PIPE_HALO = 32;                          // >= any half window, multiple of 4

struct PipeStreams { hipStream_t up = nullptr, down = nullptr; };
PipeStreams *pipe_streams(DeviceCtx *ctx)
{
    static std::mutex mu;
    static std::vector<std::pair<int, PipeStreams *>> all;
    std::lock_guard<std::mutex> lock(mu);
    for (auto &e : all) if (e.first == ctx->ordinal) return e.second;
    PipeStreams *p = new PipeStreams();
    if (hipStreamCreateWithFlags(&p->up, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&p->down, hipStreamNonBlocking) != hipSuccess) {
        delete p;
        return nullptr;
    }
    all.emplace_back(ctx->ordinal, p);
    return p;
}

// variant FULL: output[j] for j in [0, L); VALID: output[j - n] for j in [n, L - n).  Caller holds ctx->mu.  0 on success.
int host_apply_pipelined(const char *who, DeviceCtx *ctx, const SavgolFilter *f, const float *input, float *output, size_t L, Variant variant)
{
    if (!filter_sane(f, who)) return -1;
    const int n = f->config.half_window;
    PipeStreams *ps = pipe_streams(ctx);
    if (!ps) { sg_set_error("%s: could not create copy streams", who); return -1; }
    const size_t ld = (L + 3) & ~(size_t)3;
    float *d_in = static_cast<float *>(sg::ctx_arena(ctx, 2 * ld * sizeof(float)));
    if (!d_in) return -1;
    float *d_out = d_in + ld;
    const FilterPlan *plan = plan_get(ctx, f, NEED_REF);
    if (!plan) return -1;
    const float *d_table = plan->d_ref;
    const int mode = (variant == FULL) ? (int)f->config.boundary : (int)SAVGOL_BOUNDARY_POLYNOMIAL;
    const int shift = (variant == VALID) ? n : 0;
    const float dt_inv = dt_inverse(f);

    constexpr size_t PIPE_CHUNK = PIPE_CHUNK_DEFAULT;
    const size_t lo = PIPE_HALO, hi = L - PIPE_HALO;                       // centre outputs [lo, hi) go through the chunks
    const size_t nchunks = (hi - lo + PIPE_CHUNK - 1) / PIPE_CHUNK;
    std::vector<hipEvent_t> done(nchunks, nullptr);
    for (auto &e : done) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { sg_set_error("%s: hipEventCreate failed", who); return -1; }

    std::atomic<int> down_rc{0};
    std::atomic<size_t> posted{0};                                         // chunks whose event has been recorded
    std::atomic<bool> abort_flag{false};
    const int device = ctx->ordinal;
    std::thread helper([&] {
        if (hipSetDevice(device) != hipSuccess) { down_rc = -1; return; }
        for (size_t k = 0; k < nchunks; ++k) {
            while (posted.load(std::memory_order_acquire) <= k) {
                if (abort_flag.load()) return;
                std::this_thread::yield();
            }
            const size_t a = lo + k * PIPE_CHUNK, b = std::min(a + PIPE_CHUNK, hi);
            if (hipStreamWaitEvent(ps->down, done[k], 0) != hipSuccess ||
                hipMemcpyAsync(output + (a - shift), d_out + (a - shift), (b - a) * sizeof(float), hipMemcpyDeviceToHost, ps->down) != hipSuccess ||
                hipStreamSynchronize(ps->down) != hipSuccess) { down_rc = -1; return; }
        }
    });

    int rc = 0;
    size_t uploaded = 0;
    for (size_t k = 0; k < nchunks && rc == 0; ++k) {
        const size_t a = lo + k * PIPE_CHUNK, b = std::min(a + PIPE_CHUNK, hi);
        const size_t need = (k + 1 == nchunks) ? L : std::min(L, b + PIPE_HALO);          // samples this chunk's windows reach
        if (need > uploaded) {
            if (!sg::hip_ok(hipMemcpyAsync(d_in + uploaded, input + uploaded, (need - uploaded) * sizeof(float), hipMemcpyHostToDevice, ps->up), "H2D copy")) { rc = -1; break; }
            uploaded = need;
        }
        // the sub-row [a - 32, b + 32): its stored range [32, 32 + b - a) never sees what lies beyond it
        if (sg1d_launch_refpk_f32(d_in + (a - PIPE_HALO), d_out + (a - PIPE_HALO) - shift, (long long)ld, (long long)ld,
                                  (long long)((b - a) + 2 * PIPE_HALO), n, f->center_weights, dt_inv, (int)SAVGOL_BOUNDARY_CONSTANT,
                                  PIPE_HALO, PIPE_HALO + (int)(b - a), 0, 1, ctx->cu_count, ps->up) != 0) { sg_set_error("%s: kernel launch failed", who); rc = -1; break; }
        if (!sg::hip_ok(hipEventRecord(done[k], ps->up), "hipEventRecord")) { rc = -1; break; }
        posted.store(k + 1, std::memory_order_release);
    }
    if (rc != 0) abort_flag = true;
    // the 32 outputs at either end: boundary handling of every mode, reversed leading edge, VALID's narrower range
    if (rc == 0) {
        const int negate = (mode == SAVGOL_BOUNDARY_POLYNOMIAL && (g_default_flags.load() & SAVGOL_BATCH_CORRECT_LEADING_EDGE) && (f->config.derivative & 1)) ? 1 : 0;
        const int e_lo0 = (variant == VALID) ? n : 0, e_hi1 = (variant == VALID) ? (int)L - n : (int)L;
        if (sg1d_launch_reference_order_f32(d_in, d_out, (long long)ld, (long long)ld, (long long)L, n, d_table, dt_inv, mode, e_lo0, (int)lo, shift,
                                            negate, 1, ps->up) != 0 ||
            sg1d_launch_reference_order_f32(d_in, d_out, (long long)ld, (long long)ld, (long long)L, n, d_table, dt_inv, mode, (int)hi, e_hi1, shift, 0,
                                            1, ps->up) != 0) { sg_set_error("%s: edge kernel launch failed", who); rc = -1; }
        if (rc == 0 && !sg::hip_ok(hipStreamSynchronize(ps->up), who)) rc = -1;
        if (rc == 0) {
            const size_t lead = lo - (size_t)e_lo0, trail = (size_t)e_hi1 - hi;
            if (!sg::hip_ok(hipMemcpy(output + (e_lo0 - shift), d_out + (e_lo0 - shift), lead * sizeof(float), hipMemcpyDeviceToHost), "D2H copy") ||
                !sg::hip_ok(hipMemcpy(output + (hi - shift), d_out + (hi - shift), trail * sizeof(float), hipMemcpyDeviceToHost), "D2H copy")) rc = -1;
        }
    }
    helper.join();
    for (auto &e : done) (void)hipEventDestroy(e);
    if (down_rc.load() != 0) { sg_set_error("%s: D2H copy failed", who); rc = -1; }
    return rc;
}

}  // namespace

extern "C" {

int sg_option_boundary_aware(void) { return (g_default_flags.load() & SAVGOL_BATCH_BOUNDARY_AWARE) ? 1 : 0; }

unsigned savgol_hip_default_flags(void) { return g_default_flags.load(); }

int savgol_hip_set_option(int option, int value)
{
    auto set = [](unsigned bit, bool on) { if (on) g_default_flags.fetch_or(bit); else g_default_flags.fetch_and(~bit); };
    if (option == SAVGOL_HIP_OPT_CORRECT_LEADING_EDGE) { set(SAVGOL_BATCH_CORRECT_LEADING_EDGE, value != 0); return 0; }
    if (option == SAVGOL_HIP_OPT_REFERENCE_SUMMATION) { set(SAVGOL_BATCH_REFERENCE_SUMMATION, value != 0); return 0; }
    if (option == SAVGOL_HIP_OPT_PLAIN_SUMMATION) { set(SAVGOL_BATCH_PLAIN_SUMMATION, value != 0); return 0; }
    if (option == SAVGOL_HIP_OPT_BOUNDARY_AWARE) { set(SAVGOL_BATCH_BOUNDARY_AWARE, value != 0); return 0; }
    if (option == SAVGOL_HIP_OPT_TILE_WIDTH) {
        if (value < 0 || value > 2) { sg_set_error("savgol_hip_set_option: tile width %d (0 auto, 1 narrow, 2 wide)", value); return -1; }
        set(SAVGOL_BATCH_TILE_NARROW, value == 1);
        set(SAVGOL_BATCH_TILE_WIDE, value == 2);
        return 0;
    }
    sg_set_error("savgol_hip_set_option: unknown option %d", option);
    return -1;
}

int savgol_hip_momenth_table(const SavgolFilter *filter, float *table)
{
    if (!filter || !table) { sg_set_error("savgol_hip_momenth_table: NULL pointer"); return -1; }
    return sg1d_momenth_prepare(filter->config.half_window, filter->center_weights, table);
}

// Medium host signals: two hipMemcpy calls cost more than the filter.  Between these lengths the CPU copies the signal into
// pinned, device-visible host memory, the kernel reads it and writes its result across the link itself, and one stream
// synchronise ends the call: launch + synchronise instead of H2D + launch + D2H.  Same kernel, same bits.  Measured
// (tools/time_host_small.py, profiles/r03_host_small.txt): 4096 samples 30 -> 23 us, 65 536: 66 -> 53 us, 262 144: 139 -> 129 us;
// BELOW ~2000 samples it loses (360 samples, the reference's demo: 20 -> 26 us -- the kernel's dependent loads then each pay a
// PCIe round trip), so short signals keep the copies.
static bool zero_copy_length(size_t n) { return n >= 2048 && n <= 262144; }

// Short host signals (<= 4096 samples; the reference's demo filters 360): the resident small-call service of sg_k1d_misc.hip -- no
// launch, no copies through the runtime: 0 = `output` holds the result, 1 = not taken (longer signal, service disabled or
// unavailable: the caller goes on to the launched paths), -1 = error.  Caller holds ctx->mu.
extern "C" int sg_small_call(void *ctx, const float *d_table, const float *input, float *output, int L, int n, int mode, int store_lo,
                             int store_hi, int out_shift, int negate, float dt_inv);
static unsigned host_call_flags();
static int small_host_call(const char *who, DeviceCtx *ctx, const SavgolFilter *f, const float *input, float *output, size_t length, Variant variant)
{
    if (length > 4096) return 1;
    if (!filter_sane(f, who)) return -1;
    const int n = f->config.half_window;
    const FilterPlan *plan = plan_get(ctx, f, NEED_REF);
    if (!plan) return -1;
    const unsigned flags = host_call_flags();
    const int mode = (variant == FULL) ? (int)f->config.boundary : (int)SAVGOL_BOUNDARY_POLYNOMIAL;
    const bool inner = variant == VALID;
    const int negate = (mode == SAVGOL_BOUNDARY_POLYNOMIAL && (flags & SAVGOL_BATCH_CORRECT_LEADING_EDGE) && (f->config.derivative & 1)) ? 1 : 0;
    return sg_small_call(ctx, plan->d_ref, input, output, (int)length, n, mode, inner ? n : 0, inner ? (int)length - n : (int)length, inner ? n : 0, negate,
                         dt_inverse(f));
}

// the host-pointer drop-in calls: always the reference's summation order; the two semantic switches follow the process defaults
static unsigned host_call_flags()
{
    return SAVGOL_BATCH_REFERENCE_SUMMATION | (g_default_flags.load() & (SAVGOL_BATCH_CORRECT_LEADING_EDGE | SAVGOL_BATCH_BOUNDARY_AWARE));
}

static bool flags_ok(const char *who, unsigned flags)
{
    const unsigned known = SAVGOL_BATCH_REFERENCE_SUMMATION | SAVGOL_BATCH_PLAIN_SUMMATION | SAVGOL_BATCH_TILE_NARROW | SAVGOL_BATCH_TILE_WIDE |
                           SAVGOL_BATCH_CORRECT_LEADING_EDGE | SAVGOL_BATCH_BOUNDARY_AWARE | SAVGOL_BATCH_MOMENT_F64;
    if ((flags & ~known) || ((flags & SAVGOL_BATCH_TILE_NARROW) && (flags & SAVGOL_BATCH_TILE_WIDE))) {
        sg_set_error("%s: bad flags 0x%x", who, flags);
        return false;
    }
    return true;
}

int savgol_apply_batch_f32_ex(const SavgolFilter *filter, const float *d_in, float *d_out, size_t channels, size_t length,
                              size_t in_ld, size_t out_ld, unsigned flags, void *stream)
{
    if (!flags_ok("savgol_apply_batch_f32_ex", flags)) return -1;
    return enqueue_batch<float>("savgol_apply_batch_f32", filter, d_in, d_out, channels, length, in_ld, out_ld, FULL, static_cast<hipStream_t>(stream), flags);
}

int savgol_apply_batch_f64_ex(const SavgolFilter *filter, const double *d_in, double *d_out, size_t channels, size_t length,
                              size_t in_ld, size_t out_ld, unsigned flags, void *stream)
{
    if (!flags_ok("savgol_apply_batch_f64_ex", flags)) return -1;
    return enqueue_batch<double>("savgol_apply_batch_f64", filter, d_in, d_out, channels, length, in_ld, out_ld, FULL, static_cast<hipStream_t>(stream), flags);
}

// fp64 with the tolerance stated in the call (round 6): the accuracy the caller accepts picks the kernel, not a flag
// nobody sets.  rel_tol >= 1e-6 (north_star's bar for fp64 output) -> the block-moment kernel wherever the fit accepts the table (half windows
// 24..32; measured <= 1.5e-7 of the fp64 oracle); tighter -> the default tap-by-tap path (1e-12).  NaN / negative: -1.
static int batch_f64_tol(const char *who, const SavgolFilter *filter, const double *d_in, double *d_out, size_t channels, size_t length, size_t in_ld,
                         size_t out_ld, double rel_tol, Variant variant, void *stream)
{
    if (!(rel_tol >= 0.0)) { sg_set_error("%s: rel_tol must be a non-negative number", who); return -1; }
    const unsigned flags = (g_default_flags.load() & ~SAVGOL_BATCH_MOMENT_F64) | (rel_tol >= 1e-6 ? (unsigned)SAVGOL_BATCH_MOMENT_F64 : 0u);
    return enqueue_batch<double>(who, filter, d_in, d_out, channels, length, in_ld, out_ld, variant, static_cast<hipStream_t>(stream), flags);
}
int savgol_apply_batch_f64_tol(const SavgolFilter *filter, const double *d_in, double *d_out, size_t channels, size_t length, size_t in_ld, size_t out_ld,
                               double rel_tol, void *stream)
{
    return batch_f64_tol("savgol_apply_batch_f64_tol", filter, d_in, d_out, channels, length, in_ld, out_ld, rel_tol, FULL, stream);
}
int savgol_apply_valid_batch_f64_tol(const SavgolFilter *filter, const double *d_in, double *d_out, size_t channels, size_t length, size_t in_ld, size_t out_ld,
                                     double rel_tol, void *stream)
{
    return batch_f64_tol("savgol_apply_valid_batch_f64_tol", filter, d_in, d_out, channels, length, in_ld, out_ld, rel_tol, VALID, stream);
}

int savgol_apply_valid_batch_f32_ex(const SavgolFilter *filter, const float *d_in, float *d_out, size_t channels, size_t length,
                                    size_t in_ld, size_t out_ld, unsigned flags, void *stream)
{
    if (!flags_ok("savgol_apply_valid_batch_f32_ex", flags)) return -1;
    return enqueue_batch<float>("savgol_apply_valid_batch_f32", filter, d_in, d_out, channels, length, in_ld, out_ld, VALID, static_cast<hipStream_t>(stream), flags);
}

int savgol_apply_valid_batch_f64_ex(const SavgolFilter *filter, const double *d_in, double *d_out, size_t channels, size_t length,
                                    size_t in_ld, size_t out_ld, unsigned flags, void *stream)
{
    if (!flags_ok("savgol_apply_valid_batch_f64_ex", flags)) return -1;
    return enqueue_batch<double>("savgol_apply_valid_batch_f64", filter, d_in, d_out, channels, length, in_ld, out_ld, VALID, static_cast<hipStream_t>(stream), flags);
}

int savgol_apply_batch_f32(const SavgolFilter *filter, const float *d_in, float *d_out, size_t channels, size_t length,
                           size_t in_ld, size_t out_ld, void *stream)
{
    return enqueue_batch<float>("savgol_apply_batch_f32", filter, d_in, d_out, channels, length, in_ld, out_ld, FULL,
                                static_cast<hipStream_t>(stream), g_default_flags.load());
}

int savgol_apply_batch_f64(const SavgolFilter *filter, const double *d_in, double *d_out, size_t channels, size_t length,
                           size_t in_ld, size_t out_ld, void *stream)
{
    return enqueue_batch<double>("savgol_apply_batch_f64", filter, d_in, d_out, channels, length, in_ld, out_ld, FULL,
                                 static_cast<hipStream_t>(stream), g_default_flags.load());
}

int savgol_apply_valid_batch_f32(const SavgolFilter *filter, const float *d_in, float *d_out, size_t channels,
                                 size_t length, size_t in_ld, size_t out_ld, void *stream)
{
    return enqueue_batch<float>("savgol_apply_valid_batch_f32", filter, d_in, d_out, channels, length, in_ld, out_ld,
                                VALID, static_cast<hipStream_t>(stream), g_default_flags.load());
}

int savgol_apply_valid_batch_f64(const SavgolFilter *filter, const double *d_in, double *d_out, size_t channels,
                                 size_t length, size_t in_ld, size_t out_ld, void *stream)
{
    return enqueue_batch<double>("savgol_apply_valid_batch_f64", filter, d_in, d_out, channels, length, in_ld, out_ld,
                                 VALID, static_cast<hipStream_t>(stream), g_default_flags.load());
}

int savgol_apply_strided_batch_f32(const SavgolFilter *filter, const void *d_in, size_t in_stride, size_t in_offset,
                                   size_t in_channel_pitch, void *d_out, size_t out_stride, size_t out_offset,
                                   size_t out_channel_pitch, size_t channels, size_t count, void *stream)
{
    return savgol_apply_strided_batch_f32_ex(filter, d_in, in_stride, in_offset, in_channel_pitch, d_out, out_stride, out_offset,
                                             out_channel_pitch, channels, count, g_default_flags.load(), stream);
}

int savgol_apply_strided_batch_f32_ex(const SavgolFilter *filter, const void *d_in, size_t in_stride, size_t in_offset,
                                      size_t in_channel_pitch, void *d_out, size_t out_stride, size_t out_offset,
                                      size_t out_channel_pitch, size_t channels, size_t count, unsigned flags, void *stream)
{
    const char *who = "savgol_apply_strided_batch_f32";
    if (!flags_ok(who, flags)) return -1;
    if (!filter || !d_in || !d_out) { sg_set_error("%s: NULL pointer", who); return -1; }
    if (!filter_sane(filter, who)) return -1;
    if (count < (size_t)filter->window_size) { sg_set_error("%s: count < window size", who); return -1; }
    if (channels == 0) return 0;
    DeviceCtx *ctx = sg::ctx_get();
    if (!ctx) return -1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int n = filter->config.half_window;
    const Variant variant = (flags & SAVGOL_BATCH_BOUNDARY_AWARE) ? FULL : FULL_POLY_EDGES;

    // ---- one pass over the records: the field is gathered while a tile is staged and scattered when its results leave
    //      (sg1d_strided_kernel).  Needs 4-byte aligned fields (dword loads), samples a launch can index, and -- tiles read their
    //      halo while other tiles store -- input and output FIELDS that share no byte.  The same array with another field offset
    //      (the usual in-place AoS case) qualifies; the same field in place does not and takes the staged path below. ----
    const uintptr_t ia = (uintptr_t)d_in + in_offset, oa = (uintptr_t)d_out + out_offset;
    const bool aligned = ia % 4 == 0 && oa % 4 == 0 && in_stride % 4 == 0 && out_stride % 4 == 0 && in_channel_pitch % 4 == 0 && out_channel_pitch % 4 == 0 &&
                         in_stride >= 4 && out_stride >= 4;
    bool disjoint;
    {
        const uintptr_t ie = ia + (channels - 1) * in_channel_pitch + (count - 1) * in_stride + 4, oe = oa + (channels - 1) * out_channel_pitch + (count - 1) * out_stride + 4;
        if (ie <= oa || oe <= ia) disjoint = true;                          // the two arrays do not overlap at all
        else if (in_stride == out_stride && in_channel_pitch == out_channel_pitch && in_channel_pitch % in_stride == 0) {
            // one record grid: the fields are disjoint iff their offsets inside a record are at least 4 bytes apart (cyclically)
            const uintptr_t d = (ia <= oa ? oa - ia : ia - oa) % in_stride;
            disjoint = d >= 4 && in_stride - d >= 4;
        } else disjoint = false;
    }
    if (!(flags & SAVGOL_BATCH_REFERENCE_SUMMATION) && aligned && disjoint && count <= LAUNCH_MAX_LENGTH) {
        const int mode = (variant == FULL) ? (int)filter->config.boundary : (int)SAVGOL_BOUNDARY_POLYNOMIAL;
        const bool poly = mode == SAVGOL_BOUNDARY_POLYNOMIAL;
        const FilterPlan *plan = plan_get(ctx, filter, poly ? NEED_EDGES : 0u);
        if (!plan) return -1;
        sg::JobStrided job;
        memset(&job, 0, sizeof(job));
        job.in_pitch = (long long)in_channel_pitch; job.out_pitch = (long long)out_channel_pitch;
        job.in_stride = (long long)in_stride; job.out_stride = (long long)out_stride;
        job.length = (unsigned)count;
        const unsigned TW = 64u * (unsigned)SG_VPL_NARROW * 4u;
        job.tiles_per_channel = (unsigned)((count + TW - 1) / TW);
        sg::division_magic(job.tiles_per_channel, &job.tpc_magic, &job.tpc_shift);
        job.dt_inv = dt_inverse(filter);
        job.store_lo = poly ? (unsigned)n : 0u;
        job.store_hi = poly ? (unsigned)(count - n) : (unsigned)count;
        job.flags = ((unsigned)mode & sg::JOB_MODE_MASK);
        if (mode < 0 || mode > 255) job.flags = 255u;
        if (job.dt_inv != 1.0f) job.flags |= sg::JOB_SCALE;
        if (poly) {                                                      // the edge rows ride in the same launch (sg1d_edge_item)
            job.edges = plan->d_edges;
            if ((flags & SAVGOL_BATCH_CORRECT_LEADING_EDGE) && (filter->config.derivative & 1)) job.flags |= sg::JOB_EDGE_NEGATE;
        }
        const size_t max_ch = (size_t)sg::MAX_TILES_PER_LAUNCH / ((size_t)job.tiles_per_channel + 2);
        for (size_t c0 = 0; c0 < channels; c0 += max_ch) {
            const size_t nc = (channels - c0 < max_ch) ? channels - c0 : max_ch;
            job.in = reinterpret_cast<const char *>(ia) + c0 * in_channel_pitch;
            job.out = reinterpret_cast<char *>(oa) + c0 * out_channel_pitch;
            job.total_tiles = (unsigned)(nc * job.tiles_per_channel);
            job.edge_items = poly ? (unsigned)(2 * nc) : 0u;
            unsigned blocks = (job.total_tiles + job.edge_items + 3u) / 4u;
            blocks = (blocks + 7u) & ~7u;
            if (sg::launch_strided(n, job, plan->taps32, blocks, st) != 0) return -1;
        }
        return 0;
    }

    // ---- staged path (the reference's summation order, unaligned or overlapping fields, channels beyond 2^30 samples): gather the
    //      field into dense rows, filter, scatter back.  The two dense frames are this call's own, allocated and freed in stream
    //      order (sg::scratch_alloc: the library's own retained pool): no shared arena, no lock, no synchronise -- the call only enqueues. ----
    const size_t ld = (count + 3) & ~(size_t)3;
    float *dense = nullptr;
    dense = static_cast<float *>(sg::scratch_alloc(ctx, 2 * channels * ld * sizeof(float), st, "scratch (strided staging)"));
    if (!dense) return -1;
    float *result = dense + channels * ld;
    int rc = 0;
    if (sg_launch_gather_f32(d_in, in_stride, in_offset, in_channel_pitch, dense, ld, channels, count, st) != 0) {
        sg_set_error("%s: gather launch failed", who);
        rc = -1;
    }
    if (rc == 0 && enqueue_batch<float>(who, filter, dense, result, channels, count, ld, ld, variant, st, flags) != 0) rc = -1;
    if (rc == 0 && sg_launch_scatter_f32(result, ld, d_out, out_stride, out_offset, out_channel_pitch, channels, count, st) != 0) {
        sg_set_error("%s: scatter launch failed", who);
        rc = -1;
    }
    if (!sg::scratch_free(dense, st, "scratch free (strided staging)")) rc = -1;
    return rc;
}

// ------------------------------------------------------------------------------------------------
// drop-in host-pointer entry points (reference src/savgolFilter.c:743-934): stage through HBM.  They run the
// reference-order kernel (sg1d_reference_order_kernel): the host link bounds them anyway, and a program that switches
// from the reference library gets the reference's outputs bit for bit.
// ------------------------------------------------------------------------------------------------
int savgol_apply(const SavgolFilter *filter, const float *input, float *output, size_t length)
{
    if (!filter || !input || !output) {
        fprintf(stderr, "savgol_apply: NULL pointer\n");
        return -1;
    }
    if (length < (size_t)filter->window_size) {
        fprintf(stderr, "savgol_apply: data length (%lu) < window size (%d)\n", (unsigned long)length, filter->window_size);
        return -1;
    }
    DeviceCtx *ctx = sg::ctx_get();
    if (!ctx) { fprintf(stderr, "savgol_apply: %s\n", savgol_hip_last_error()); return -1; }
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    {
        const int rc = small_host_call("savgol_apply", ctx, filter, input, output, length, FULL);
        if (rc == 0) return 0;
        if (rc < 0) { fprintf(stderr, "savgol_apply: %s\n", savgol_hip_last_error()); return -1; }
    }
    // pipelined unless the two host buffers overlap without being the same (in place is fine: a chunk is downloaded only
    // after the samples it overwrites went up; a shifted overlap is not, so that case keeps the upload-everything-first path)
    const bool partial_overlap = input != output && (uintptr_t)input < (uintptr_t)(output + length) && (uintptr_t)output < (uintptr_t)(input + length);
    if (length >= PIPE_MIN_LENGTH && length <= ((size_t)1 << 30) && !partial_overlap) {
        if (host_apply_pipelined("savgol_apply", ctx, filter, input, output, length, FULL) == 0) return 0;
        fprintf(stderr, "savgol_apply: %s\n", savgol_hip_last_error());
        return -1;
    }
    const size_t ld = (length + 3) & ~(size_t)3;
    if (zero_copy_length(length)) {
        float *pin = static_cast<float *>(sg::ctx_pinned(ctx, 2 * ld * sizeof(float)));
        if (!pin) { fprintf(stderr, "savgol_apply: %s\n", savgol_hip_last_error()); return -1; }
        memcpy(pin, input, length * sizeof(float));
        if (enqueue_batch<float>("savgol_apply", filter, pin, pin + ld, 1, length, ld, ld, FULL, nullptr, host_call_flags()) != 0 ||
            !sg::hip_ok(hipStreamSynchronize(nullptr), "savgol_apply")) { fprintf(stderr, "savgol_apply: %s\n", savgol_hip_last_error()); return -1; }
        memcpy(output, pin + ld, length * sizeof(float));
        return 0;
    }
    float *d_in = static_cast<float *>(sg::ctx_arena(ctx, 2 * ld * sizeof(float)));
    if (!d_in) { fprintf(stderr, "savgol_apply: %s\n", savgol_hip_last_error()); return -1; }
    float *d_out = d_in + ld;
    bool ok = sg::hip_ok(hipMemcpyAsync(d_in, input, length * sizeof(float), hipMemcpyHostToDevice, nullptr), "H2D copy");
    ok = ok && enqueue_batch<float>("savgol_apply", filter, d_in, d_out, 1, length, ld, ld, FULL, nullptr, host_call_flags()) == 0;
    ok = ok && sg::hip_ok(hipMemcpy(output, d_out, length * sizeof(float), hipMemcpyDeviceToHost), "D2H copy");
    if (!ok) { fprintf(stderr, "savgol_apply: %s\n", savgol_hip_last_error()); return -1; }
    return 0;
}

size_t savgol_apply_valid(const SavgolFilter *filter, const float *input, size_t input_length, float *output)
{
    if (!filter || !input || !output) return 0;
    if (input_length < (size_t)filter->window_size) return 0;
    DeviceCtx *ctx = sg::ctx_get();
    if (!ctx) { fprintf(stderr, "savgol_apply_valid: %s\n", savgol_hip_last_error()); return 0; }
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    const size_t out_len = input_length - 2 * (size_t)filter->config.half_window;
    {
        const int rc = small_host_call("savgol_apply_valid", ctx, filter, input, output, input_length, VALID);
        if (rc == 0) return out_len;
        if (rc < 0) { fprintf(stderr, "savgol_apply_valid: %s\n", savgol_hip_last_error()); return 0; }
    }
    // VALID writes output[j - n]: even output == input is a shifted overlap, so only disjoint buffers are pipelined
    const bool overlap = (uintptr_t)input < (uintptr_t)(output + out_len) && (uintptr_t)output < (uintptr_t)(input + input_length);
    if (input_length >= PIPE_MIN_LENGTH && input_length <= ((size_t)1 << 30) && !overlap) {
        if (host_apply_pipelined("savgol_apply_valid", ctx, filter, input, output, input_length, VALID) == 0) return out_len;
        fprintf(stderr, "savgol_apply_valid: %s\n", savgol_hip_last_error());
        return 0;
    }
    const size_t ld = (input_length + 3) & ~(size_t)3;
    if (zero_copy_length(input_length)) {
        float *pin = static_cast<float *>(sg::ctx_pinned(ctx, 2 * ld * sizeof(float)));
        if (!pin) { fprintf(stderr, "savgol_apply_valid: %s\n", savgol_hip_last_error()); return 0; }
        memcpy(pin, input, input_length * sizeof(float));
        if (enqueue_batch<float>("savgol_apply_valid", filter, pin, pin + ld, 1, input_length, ld, ld, VALID, nullptr, host_call_flags()) != 0 ||
            !sg::hip_ok(hipStreamSynchronize(nullptr), "savgol_apply_valid")) { fprintf(stderr, "savgol_apply_valid: %s\n", savgol_hip_last_error()); return 0; }
        memcpy(output, pin + ld, out_len * sizeof(float));
        return out_len;
    }
    float *d_in = static_cast<float *>(sg::ctx_arena(ctx, 2 * ld * sizeof(float)));
    if (!d_in) { fprintf(stderr, "savgol_apply_valid: %s\n", savgol_hip_last_error()); return 0; }
    float *d_out = d_in + ld;
    bool ok = sg::hip_ok(hipMemcpyAsync(d_in, input, input_length * sizeof(float), hipMemcpyHostToDevice, nullptr), "H2D copy");
    ok = ok && enqueue_batch<float>("savgol_apply_valid", filter, d_in, d_out, 1, input_length, ld, ld, VALID, nullptr, host_call_flags()) == 0;
    ok = ok && sg::hip_ok(hipMemcpy(output, d_out, out_len * sizeof(float), hipMemcpyDeviceToHost), "D2H copy");
    if (!ok) { fprintf(stderr, "savgol_apply_valid: %s\n", savgol_hip_last_error()); return 0; }
    return out_len;
}

int savgol_apply_strided(const SavgolFilter *filter, const void *input, size_t in_stride, size_t in_offset, void *output,
                         size_t out_stride, size_t out_offset, size_t count)
{
    if (!filter || !input || !output) return -1;
    if (count < (size_t)filter->window_size) return -1;
    DeviceCtx *ctx = sg::ctx_get();
    if (!ctx) { fprintf(stderr, "savgol_apply_strided: %s\n", savgol_hip_last_error()); return -1; }
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    // the float fields are picked on the host (pure data movement), the arithmetic runs on the GPU
    const size_t ld = (count + 3) & ~(size_t)3;
    float *stage = static_cast<float *>(sg::ctx_pinned(ctx, 2 * ld * sizeof(float)));      // gathered field, then (short signals) the result
    float *d_in = static_cast<float *>(sg::ctx_arena(ctx, 2 * ld * sizeof(float)));
    if (!stage || !d_in) { fprintf(stderr, "savgol_apply_strided: %s\n", savgol_hip_last_error()); return -1; }
    float *d_out = d_in + ld;
    const char *ib = static_cast<const char *>(input) + in_offset;
    for (size_t i = 0; i < count; ++i) memcpy(&stage[i], ib + i * in_stride, sizeof(float));
    {
        const int rc = small_host_call("savgol_apply_strided", ctx, filter, stage, stage + ld, count, (host_call_flags() & SAVGOL_BATCH_BOUNDARY_AWARE) ? FULL : FULL_POLY_EDGES);
        if (rc < 0) { fprintf(stderr, "savgol_apply_strided: %s\n", savgol_hip_last_error()); return -1; }
        if (rc == 0) {
            char *ob0 = static_cast<char *>(output) + out_offset;
            for (size_t i = 0; i < count; ++i) memcpy(ob0 + i * out_stride, &stage[ld + i], sizeof(float));
            return 0;
        }
    }
    if (zero_copy_length(count)) {                               // short signals: the kernel works on the pinned staging buffer itself
        float *pin = stage;
        if (enqueue_batch<float>("savgol_apply_strided", filter, pin, pin + ld, 1, count, ld, ld, (host_call_flags() & SAVGOL_BATCH_BOUNDARY_AWARE) ? FULL : FULL_POLY_EDGES,
                                 nullptr, host_call_flags()) != 0 ||
            !sg::hip_ok(hipStreamSynchronize(nullptr), "savgol_apply_strided")) { fprintf(stderr, "savgol_apply_strided: %s\n", savgol_hip_last_error()); return -1; }
        char *ob2 = static_cast<char *>(output) + out_offset;
        for (size_t i = 0; i < count; ++i) memcpy(ob2 + i * out_stride, &pin[ld + i], sizeof(float));
        return 0;
    }
    bool ok = sg::hip_ok(hipMemcpy(d_in, stage, count * sizeof(float), hipMemcpyHostToDevice), "H2D copy");
    ok = ok && enqueue_batch<float>("savgol_apply_strided", filter, d_in, d_out, 1, count, ld, ld, (host_call_flags() & SAVGOL_BATCH_BOUNDARY_AWARE) ? FULL : FULL_POLY_EDGES, nullptr, host_call_flags()) == 0;
    ok = ok && sg::hip_ok(hipMemcpy(stage, d_out, count * sizeof(float), hipMemcpyDeviceToHost), "D2H copy");
    if (!ok) { fprintf(stderr, "savgol_apply_strided: %s\n", savgol_hip_last_error()); return -1; }
    char *ob = static_cast<char *>(output) + out_offset;
    for (size_t i = 0; i < count; ++i) memcpy(ob + i * out_stride, &stage[i], sizeof(float));
    return 0;
}

}  // extern "C"
