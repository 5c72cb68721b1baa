/*
 * sg_oracle.h -- CPU oracle for the Savitzky-Golay hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (libsavgol_hip.so) never links, loads or calls anything declared here.
 *
 * Every function is a plain-C restatement of one piece of the reference
 * (Tugbars/Savitzky-Golay-Filter); the reference file:line each one follows is given in
 * sg_oracle.c.  Parity is PINNED: tests/test_oracle_pinned.py checks this restatement
 *   (1) bit-for-bit against the compiled, unmodified reference (oracle/_ref/libsavgol_ref.so,
 *       built by `make -C oracle ref`) whenever that library is present, and
 *   (2) bit-for-bit against the committed golden fixtures in tests/golden/ (which were produced
 *       by that same compiled reference, see tests/golden/make_golden.py), and
 *   (3) against the only golden vector the reference itself ships (the 301-point MATLAB
 *       comparison pair in "tool for matlab comparisons/savgolComparison.m").
 */
#ifndef SG_ORACLE_H
#define SG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGO_MAX_N   32
#define SGO_MAX_WS  (2 * SGO_MAX_N + 1)

/* boundary codes = the reference's SavgolBoundaryMode values (savgolFilter.h:63-68) */
enum { SGO_POLYNOMIAL = 0, SGO_REFLECT = 1, SGO_PERIODIC = 2, SGO_CONSTANT = 3 };
/* 2-D boundary codes = Savgol2DBoundary (savgol2d.h:108-112) */
enum { SGO2D_VALID = 0, SGO2D_CONSTANT = 1, SGO2D_REFLECT = 2 };

/* ---- weights (fp32 tables, same operation order as the reference) ---- */
int   sgo_weights(int n, int m, int d, float *center /*[2n+1]*/, float *edges /*[n][2n+1] packed*/);
float sgo_dt_scale(float time_step, int d);          /* powf(time_step, d)            */
float sgo_dt_inv(float time_step, int d);            /* dt_scale != 0 ? 1/dt_scale : 1 */

/* ---- 1-D batch, fp32, reference summation order (bit-exact restatement) ---- */
int    sgo_apply_f32(const float *center, const float *edges, int n, float dt_inv, int mode,
                     const float *in, float *out, size_t length);
size_t sgo_apply_valid_f32(const float *center, int n, float dt_inv,
                           const float *in, size_t length, float *out);
int    sgo_apply_strided_f32(const float *center, const float *edges, int n, float dt_inv,
                             const void *in, size_t in_stride, size_t in_offset,
                             void *out, size_t out_stride, size_t out_offset, size_t count);
/* many channels, row-major [channels][ld]; OpenMP over channels when threads > 1 */
int    sgo_apply_batch_f32(const float *center, const float *edges, int n, float dt_inv, int mode,
                           const float *in, float *out, size_t channels, size_t length, size_t ld,
                           int threads);

/* ---- 1-D batch, fp64 oracle: fp32 tables promoted exactly, double accumulation ---- */
int    sgo_apply_f64(const float *center, const float *edges, int n, float dt_inv, int mode,
                     const double *in, double *out, size_t length);
int    sgo_apply_batch_f64(const float *center, const float *edges, int n, float dt_inv, int mode,
                           const double *in, double *out, size_t channels, size_t length, size_t ld,
                           int threads);

/* ---- streaming (ring buffer + per-sample dot product) ---- */
typedef struct {
    float    ring[SGO_MAX_WS];
    int      wp;
    uint64_t received;
    uint64_t emitted;
} SgoStream;

void  sgo_stream_reset(SgoStream *s);
/* returns 1 and stores *y when an output exists, else 0 */
int   sgo_stream_push(SgoStream *s, const float *center, int n, float dt_inv, float x, float *y);
int   sgo_stream_push_full(SgoStream *s, const float *center, const float *edges, int n,
                           float dt_inv, float x, float *out, int max_out);
int   sgo_stream_flush(SgoStream *s, const float *edges, int n, float dt_inv, float *out, int max_out);
int   sgo_stream_flush_leading(SgoStream *s, const float *edges, int n, float dt_inv,
                               float *out, int max_out);

/* ---- 2-D ---- */
int   sgo2d_weights(int nx, int ny, int order, int dx, int dy, float *W /*[2ny+1][2nx+1]*/);
float sgo2d_scale(float delta_x, float delta_y, int dx, int dy);
int   sgo2d_apply_valid_f32(const float *W, int nx, int ny, float scale,
                            const float *in, int rows, int cols, int in_stride,
                            float *out, int out_stride);
int   sgo2d_apply_f32(const float *W, int nx, int ny, float scale,
                      const float *in, int rows, int cols, int in_stride,
                      float *out, int out_stride, int boundary);
/* double-accumulation variant of the same two (fp32 W promoted exactly) */
int   sgo2d_apply_f64acc(const float *W, int nx, int ny, float scale,
                         const float *in, int rows, int cols, int in_stride,
                         double *out, int out_stride, int boundary);

/* ---- synthetic workload generator of SURVEY.md section 8(d) (host version) ---- */
void  sgo_synth_f32(float *dst, size_t channel0, size_t channels, size_t length, size_t ld, uint64_t seed);
void  sgo_synth_f64(double *dst, size_t channel0, size_t channels, size_t length, size_t ld, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
