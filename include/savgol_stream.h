/*
 * savgol_stream.h -- sample-at-a-time Savitzky-Golay filtering (one stream, host API).
 *
 * Drop-in for the reference's include/iterative/savgol_stream.h: same SavgolStream POD (296 bytes,
 * fields at 0/8/268/272/280/288/292; callers may allocate it themselves for savgol_stream_init)
 * and the same 13 entry points (:49-126).  The ring buffer lives in the caller-visible struct
 * exactly as in the reference; the per-sample dot products run in a HIP kernel (no CPU fallback).
 * For many concurrent streams use the stream bank in savgol_hip.h, which keeps the rings in HBM
 * and advances all of them with one launch per tick.
 *
 * Semantics (reference src/savgol_stream.c): nothing is produced until 2n+1 samples arrived;
 * group delay is n samples; edges always use the polynomial rows; single-accumulator summation
 * in tap order, so results are bit-identical to the reference's.
 */
#ifndef SAVGOL_STREAM_H
#define SAVGOL_STREAM_H

#include <stdbool.h>
#include <stddef.h>
#include "savgolFilter.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SavgolStream {
    const SavgolFilter *filter;        /* weights (owned iff owns_filter)            */
    float  buffer[SAVGOL_MAX_WINDOW];  /* ring of the last 2n+1 samples              */
    int    write_pos;                  /* next slot to overwrite = oldest sample     */
    size_t samples_received;
    size_t samples_output;
    bool   owns_filter;
    float  dt_inv;                     /* 1 / dt_scale                               */
} SavgolStream;

SavgolStream *savgol_stream_create(const SavgolConfig *config);                  /* :49  */
int    savgol_stream_init(SavgolStream *stream, const SavgolFilter *filter);      /* :58  */
void   savgol_stream_destroy(SavgolStream *stream);                               /* :64  */
void   savgol_stream_reset(SavgolStream *stream);                                 /* :70  */

/* Centre outputs only; *output_valid (may be NULL) tells whether the return value is one. */
float  savgol_stream_push(SavgolStream *stream, float sample, bool *output_valid);           /* :84 */
/* With edges: the push that fills the window yields n leading-edge values + 1 centre value
 * (truncated to max_outputs), later pushes yield 1.  Returns the count written. */
int    savgol_stream_push_full(SavgolStream *stream, float sample, float *output, int max_outputs); /* :95 */
/* Trailing edge: up to n values (rows n-1 .. 0).  -1 on bad arguments, 0 if never filled. */
int    savgol_stream_flush(SavgolStream *stream, float *output, int max_count);              /* :106 */
int    savgol_stream_flush_leading(SavgolStream *stream, float *output, int max_count);      /* :116 */

bool   savgol_stream_ready(const SavgolStream *stream);                           /* :122 */
size_t savgol_stream_latency(const SavgolStream *stream);                         /* :123 */
size_t savgol_stream_buffered(const SavgolStream *stream);                        /* :124 */
size_t savgol_stream_samples_received(const SavgolStream *stream);                /* :125 */
size_t savgol_stream_samples_output(const SavgolStream *stream);                  /* :126 */

#ifdef __cplusplus
}
#endif
#endif /* SAVGOL_STREAM_H */
