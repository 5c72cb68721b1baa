// sg_stream.hpp -- shared by the streaming kernel files.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

#include "sg_internal.h"

// the bank behind the opaque SavgolStreamBank of savgol_hip.h (sg_stream.hip owns it; sg_stream_service.hip adds the resident
// tick service)
struct SavgolStreamBank {
    SavgolFilter *filter;
    size_t        streams;
    int           device;
    float        *d_ring;            // [ws][streams]
    const float  *d_table;           // [n+1][ws], or [2n+1][ws] with boundary-aware edges (filter_table, sg_stream.hip)
    int           trail_base;        // first trailing-edge row of d_table
    int           wp;
    unsigned long long received, emitted;
    float         dt_inv;
    void         *service;           // sg::BankService while savgol_streambank_service_* is active, else NULL
    unsigned      flags;             // SAVGOL_STREAMBANK_* of savgol_streambank_create_ex
    // savgol_streambank_push_wait: a completion word in pinned host memory that the tick kernel's last block writes and the host spins on --
    // instead of hipStreamSynchronize's several microseconds.  0 = not tried yet, 1 = in use, -1 = unavailable (push + synchronise).
    volatile unsigned *signal;       // host view
    unsigned     *signal_dev;        // device view of the same word
    unsigned     *signal_counter;    // device word: blocks of the running tick that have finished
    int           signal_state;
    unsigned      signal_seq;
};

namespace sg {

constexpr int STREAM_ROLL_MAX_N = 32;   // the rolling block-push kernel covers every half window ...
#ifndef STREAM_RING_MAX_N
#define STREAM_RING_MAX_N 16             // ... sample ring (tick loop unrolled 2n+4 times) up to here, accumulator ring above
#endif

// sg_stream_roll.hip: `ticks` pushes of every stream in one launch, outputs only (the caller updates the ring).
// 0 = launched, 1 = not covered (ticks >= 2^31): use the LDS-tiled kernel of sg_stream.hip.
int sg_bank_roll_launch(int n, const float *center_weights, const float *ring, const float *samples, float *out, size_t streams,
                        int wp0, unsigned long long received0, size_t ticks, float dt_inv, int fma, int cu_count, hipStream_t st);

}  // namespace sg
