/*
 * savgolFilter.h -- 1-D Savitzky-Golay filter, MI355X (gfx950) implementation.
 *
 * Drop-in for the reference's include/iterative/savgolFilter.h: identical type names, struct
 * layouts (callers read the fields), constants, macros and the five entry points.  Weight tables
 * are generated on the host bit-for-bit as the reference does; every savgol_apply* call runs the
 * sliding-window convolution in hand-written HIP kernels (there is no CPU fallback: without a
 * usable HIP device the apply functions fail with -1 / 0 and a message on stderr).
 *
 * Reference interface replaced, by line of include/iterative/savgolFilter.h:
 *   constants :39-48, SavgolBoundaryMode :63-68, SavgolConfig :92-98, SavgolFilter :107-113,
 *   savgol_create :130, savgol_destroy :137, savgol_apply :152, savgol_apply_strided :181-184,
 *   savgol_apply_valid :201-203, SAVGOL_SMOOTH / SAVGOL_DERIV1 / SAVGOL_DERIV2 :210-222.
 *
 * Device-pointer / batched / fp64 entry points live in savgol_hip.h.
 */
#ifndef SAVGOL_FILTER_H
#define SAVGOL_FILTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAVGOL_MAX_HALF_WINDOW 32
#define SAVGOL_MAX_WINDOW      (2 * SAVGOL_MAX_HALF_WINDOW + 1)
#define SAVGOL_MAX_POLY_ORDER  10
#define SAVGOL_MAX_DERIVATIVE  4

/* How the first/last half_window samples are produced. */
typedef enum {
    SAVGOL_BOUNDARY_POLYNOMIAL = 0, /* asymmetric least-squares rows (edge_weights)          */
    SAVGOL_BOUNDARY_REFLECT,        /* half-sample mirror:  d1 d0 | d0 d1 d2 ...             */
    SAVGOL_BOUNDARY_PERIODIC,       /* wrap-around                                           */
    SAVGOL_BOUNDARY_CONSTANT        /* repeat the end sample                                 */
} SavgolBoundaryMode;

/* 12 bytes, fields at 0/1/2/4/8. */
typedef struct {
    uint8_t half_window;            /* n, window = 2n+1, 1..SAVGOL_MAX_HALF_WINDOW           */
    uint8_t poly_order;             /* m < 2n+1 (see savgol_create for the table bound)      */
    uint8_t derivative;             /* d <= min(m, SAVGOL_MAX_DERIVATIVE); 0 = smoothing     */
    float   time_step;              /* sample spacing, > 0; outputs are divided by dt^d      */
    SavgolBoundaryMode boundary;
} SavgolConfig;

/* 8600 bytes, fields at 0/12/16/20/280.  Read-only after savgol_create(). */
typedef struct SavgolFilter {
    SavgolConfig config;
    int   window_size;                                              /* 2n+1                  */
    float dt_scale;                                                 /* time_step^derivative  */
    float center_weights[SAVGOL_MAX_WINDOW];                        /* taps for the interior */
    float edge_weights[SAVGOL_MAX_HALF_WINDOW][SAVGOL_MAX_WINDOW];  /* row e: e samples from the end */
} SavgolFilter;

/* Validate, allocate, fill the tables (host only).  NULL on a bad config or allocation failure.
 * One deliberate deviation: configs with 2n + m + 1 >= 76 (only possible for m > 10 at large n)
 * are rejected -- the reference accepts them, indexes past its 76-entry factorial table, prints
 * "GenFact lookup out of range" and returns garbage weights (src/savgolFilter.c:110,187-192). */
SavgolFilter *savgol_create(const SavgolConfig *config);
void          savgol_destroy(SavgolFilter *filter);           /* NULL is a no-op */

/* Host buffers in, host buffers out; the convolution runs on the GPU.
 * length >= window_size, else -1.  output may alias input (the result is the out-of-place one). */
int savgol_apply(const SavgolFilter *filter, const float *input, float *output, size_t length);

/* Array-of-structs variant: element i is the float at (char*)base + i*stride + offset.
 * Edges always use the polynomial rows, whatever config.boundary says (as the reference does). */
int savgol_apply_strided(const SavgolFilter *filter,
                         const void *input, size_t in_stride, size_t in_offset,
                         void *output, size_t out_stride, size_t out_offset,
                         size_t count);

/* Only the samples whose whole window exists: writes input_length - 2n values, returns that
 * count (0 on error). */
size_t savgol_apply_valid(const SavgolFilter *filter,
                          const float *input, size_t input_length, float *output);

#define SAVGOL_SMOOTH(half_win, order) \
    (SavgolConfig){ .half_window = (half_win), .poly_order = (order), .derivative = 0, \
                    .time_step = 1.0f, .boundary = SAVGOL_BOUNDARY_POLYNOMIAL }
#define SAVGOL_DERIV1(half_win, order, dt) \
    (SavgolConfig){ .half_window = (half_win), .poly_order = (order), .derivative = 1, \
                    .time_step = (dt), .boundary = SAVGOL_BOUNDARY_POLYNOMIAL }
#define SAVGOL_DERIV2(half_win, order, dt) \
    (SavgolConfig){ .half_window = (half_win), .poly_order = (order), .derivative = 2, \
                    .time_step = (dt), .boundary = SAVGOL_BOUNDARY_POLYNOMIAL }

#ifdef __cplusplus
}
#endif
#endif /* SAVGOL_FILTER_H */
