// sg_runtime.hpp -- process-wide HIP plumbing shared by the host-side sources: per-device context
// (scratch arena for the host-pointer drop-in calls, cache of uploaded weight tables), error text.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <mutex>
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
    std::vector<TableEntry> tables;
};

// Context of the device this thread uses (savgol_hip_set_device, else the current HIP device).
// nullptr + error text when there is no usable device.
DeviceCtx *ctx_get();

// Device copy of a host float table, uploaded once per distinct content (synchronous on first use).
const float *ctx_table(DeviceCtx *ctx, const void *host, size_t bytes, uint64_t salt);

// Scratch of at least `bytes`; caller holds ctx->mu for as long as it uses the memory.
void *ctx_arena(DeviceCtx *ctx, size_t bytes);
void *ctx_pinned(DeviceCtx *ctx, size_t bytes);

bool hip_ok(hipError_t e, const char *what);      // false + error text on failure

}  // namespace sg
