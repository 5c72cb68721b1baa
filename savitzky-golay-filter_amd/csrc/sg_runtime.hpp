// sg_runtime.hpp -- process-wide HIP plumbing shared by the host-side sources: per-device context
// (scratch arena for the host-pointer drop-in calls, cache of uploaded weight tables), error text.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "sg_internal.h"

namespace sg {

struct TableEntry {            // one uploaded float table (edge rows, 2-D kernels, separable factors); lives as long as the process
    uint64_t key;
    size_t   bytes;
    float   *dev;
    std::vector<unsigned char> host;    // the content, compared on a key hit
};

struct DeviceCtx {
    int                     ordinal = -1;
    int                     cu_count = 0;
    std::recursive_mutex    mu;              // guards everything below
    void                   *arena = nullptr; // scratch for host-pointer calls
    size_t                  arena_bytes = 0;
    void                   *pinned = nullptr; // small pinned buffer for scalar results
    size_t                  pinned_bytes = 0;
    std::unordered_multimap<uint64_t, TableEntry> tables;    // by content hash
    hipMemPool_t            pool = nullptr;  // the library's own stream-ordered pool (scratch_alloc)
    void                   *small = nullptr;  // sg::SmallService (sg_k1d_misc.hip): the resident kernel behind short host-pointer calls
};

// 64-bit content hash, eight bytes at a time on four independent lanes (a table is hashed on every call that uses it: the
// byte-wise FNV of rounds 1-2 cost ~9 us per 8.6 KB edge table, this ~0.2 us)
inline uint64_t hash64(const void *p, size_t n, uint64_t seed)
{
    const unsigned char *b = static_cast<const unsigned char *>(p);
    uint64_t h[4] = {seed ^ 0x9E3779B97F4A7C15ull, seed + 0xBF58476D1CE4E5B9ull, seed ^ 0x94D049BB133111EBull, seed + 0xD6E8FEB86659FD93ull};
    auto mix = [](uint64_t a, uint64_t w) { a ^= w; a *= 0xFF51AFD7ED558CCDull; return (a << 29) | (a >> 35); };
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        memcpy(w, b + i, 32);
        h[0] = mix(h[0], w[0]); h[1] = mix(h[1], w[1]); h[2] = mix(h[2], w[2]); h[3] = mix(h[3], w[3]);
    }
    for (int l = 0; i < n; i += 8, ++l) {
        uint64_t w = 0;
        memcpy(&w, b + i, n - i < 8 ? n - i : 8);
        h[l & 3] = mix(h[l & 3], w);
    }
    uint64_t r = (h[0] ^ (h[1] << 1 | h[1] >> 63)) + (h[2] ^ (h[3] << 7 | h[3] >> 57)) + n;
    r ^= r >> 33; r *= 0xC4CEB9FE1A85EC53ull; r ^= r >> 29;
    return r;
}

// Context of the device this thread uses (savgol_hip_set_device, else the current HIP device).
// nullptr + error text when there is no usable device.
DeviceCtx *ctx_get();

// Device copy of a host float table, uploaded once per distinct content (synchronous on first use).
const float *ctx_table(DeviceCtx *ctx, const void *host, size_t bytes, uint64_t salt);

// Scratch of at least `bytes`; caller holds ctx->mu for as long as it uses the memory.
void *ctx_arena(DeviceCtx *ctx, size_t bytes);
void *ctx_pinned(DeviceCtx *ctx, size_t bytes);

// Stream-ordered scratch for the calls that need a temporary frame (strided staging, channel ends of very long channels, row-band
// strips): allocated and freed in the order of `st`, so such a call only enqueues.  From the library's OWN memory pool: the DEFAULT
// pool on the legacy NULL stream hands out memory that queued kernels still use -- tools/repro_null_stream_pool.hip reproduces it
// (profiles/r04_null_stream_pool_repro.txt: 213 632 wrong words in 400 allocate / fill / consume / free rounds; never on a created
// stream, never with a private pool at ANY release threshold).  Round 3 had blamed the pool's release and pinned the threshold to
// UINT64_MAX, which would hide GiBs of peak scratch from the caller's allocator; it is the default pool that is broken, not releasing,
// so the threshold is 256 MiB now (SAVGOL_HIP_SCRATCH_KEEP_MB) and scratch_trim() / savgol_hip_trim_scratch() /
// savgol_hip_synchronize() hand back the rest.
uint64_t scratch_keep_bytes();                    // the pool's release threshold: 256 MiB, or SAVGOL_HIP_SCRATCH_KEEP_MB
void *scratch_alloc(DeviceCtx *ctx, size_t bytes, hipStream_t st, const char *what);   // nullptr + error text on failure
bool scratch_free(void *p, hipStream_t st, const char *what);
int scratch_trim(DeviceCtx *ctx);                 // 0 / -1

bool hip_ok(hipError_t e, const char *what);      // false + error text on failure

}  // namespace sg
