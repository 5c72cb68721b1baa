/*
 * c_api_demo.c -- plain C host code driving the HIP kernels through the C ABI, the way a user of the reference
 * library would after switching to libsavgol_hip.so.
 *
 *   make -C savitzky-golay-filter_amd examples     ->  savitzky-golay-filter_amd/lib/c_api_demo
 *
 * 1. drop-in call: savgol_create / savgol_apply on host buffers (the reference's own API);
 * 2. device-resident batch: hipMalloc'd channels, savgol_apply_batch_f32, cross-checked against (1);
 * 3. stream bank: one launch per tick, wall-clock latency per tick measured here in C (p50 / p99).
 * Exit code 0 = all checks passed.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <hip/hip_runtime_api.h>

#include "savgolFilter.h"
#include "savgol_hip.h"

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "FAILED: %s (%s)\n", #x, savgol_hip_last_error()); return 1; } } while (0)

static double now_us(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

static int cmp_double(const void *a, const void *b) { return (*(const double *)a > *(const double *)b) - (*(const double *)a < *(const double *)b); }

int main(void)
{
    CHECK(savgol_hip_device_count() > 0);

    /* ---- 1. the reference API, unchanged source ---- */
    const size_t L = 100000, CH = 64;
    SavgolConfig cfg = SAVGOL_SMOOTH(32, 4);
    SavgolFilter *f = savgol_create(&cfg);
    CHECK(f != NULL);
    float *x = malloc(sizeof(float) * L * CH), *y = malloc(sizeof(float) * L), *yb = malloc(sizeof(float) * L * CH);
    for (size_t i = 0; i < L * CH; ++i) x[i] = sinf(0.001f * (float)(i % L)) + 0.1f * (float)rand() / (float)RAND_MAX;
    CHECK(savgol_apply(f, x + 5 * L, y, L) == 0);               /* channel 5 through the drop-in entry point */

    /* ---- 2. the same filter on channels that already live in HBM ---- */
    float *d_in, *d_out;
    CHECK(hipMalloc((void **)&d_in, sizeof(float) * L * CH) == hipSuccess);
    CHECK(hipMalloc((void **)&d_out, sizeof(float) * L * CH) == hipSuccess);
    CHECK(hipMemcpy(d_in, x, sizeof(float) * L * CH, hipMemcpyHostToDevice) == hipSuccess);
    CHECK(savgol_apply_batch_f32(f, d_in, d_out, CH, L, L, L, NULL) == 0);
    CHECK(hipMemcpy(yb, d_out, sizeof(float) * L * CH, hipMemcpyDeviceToHost) == hipSuccess);
    /* the drop-in call sums in the reference's order, the batch kernel with FMAs: they agree to fp32 rounding ... */
    double apart = 0.0;
    for (size_t i = 0; i < L; ++i) { double d = fabs((double)yb[5 * L + i] - (double)y[i]); if (d > apart) apart = d; }
    CHECK(apart < 2e-6);
    /* ... and bit for bit when the batch call is asked for the reference's order too */
    CHECK(savgol_hip_set_option(SAVGOL_HIP_OPT_REFERENCE_SUMMATION, 1) == 0);
    CHECK(savgol_apply_batch_f32(f, d_in, d_out, CH, L, L, L, NULL) == 0);
    CHECK(savgol_hip_set_option(SAVGOL_HIP_OPT_REFERENCE_SUMMATION, 0) == 0);
    CHECK(hipMemcpy(yb, d_out, sizeof(float) * L * CH, hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(memcmp(yb + 5 * L, y, sizeof(float) * L) == 0);
    double worst = 0.0;                                          /* smoothing a smooth signal: stays close to it */
    for (size_t i = 100; i < L - 100; ++i) { double d = fabs(yb[i] - sinf(0.001f * (float)i) - 0.05); if (d > worst) worst = d; }
    CHECK(worst < 0.05);
    printf("batch: %zu channels x %zu samples filtered on the GPU; drop-in and device-resident results %.1e apart, identical in reference-order mode\n", CH, L, apart);

    /* ---- 3. 65 536 concurrent streams, one tick per launch ---- */
    const size_t S = 65536;
    SavgolConfig scfg = SAVGOL_DERIV1(16, 2, 1e-3f);
    SavgolStreamBank *bank = savgol_streambank_create(&scfg, S);
    CHECK(bank != NULL);
    float *d_s, *d_o;
    CHECK(hipMalloc((void **)&d_s, sizeof(float) * S) == hipSuccess);
    CHECK(hipMalloc((void **)&d_o, sizeof(float) * S) == hipSuccess);
    CHECK(hipMemset(d_s, 0, sizeof(float) * S) == hipSuccess);
    enum { TICKS = 2000 };
    static double lat[TICKS];
    for (int t = 0; t < 100; ++t) CHECK(savgol_streambank_push(bank, d_s, d_o, NULL) >= 0);
    CHECK(savgol_hip_synchronize(NULL) == 0);
    for (int t = 0; t < TICKS; ++t) {
        const double t0 = now_us();
        CHECK(savgol_streambank_push(bank, d_s, d_o, NULL) == 1);
        CHECK(savgol_hip_synchronize(NULL) == 0);
        lat[t] = now_us() - t0;
    }
    qsort(lat, TICKS, sizeof(double), cmp_double);
    printf("stream bank: %zu streams, n=16 m=2 d=1: per-tick wall latency p50 %.1f us  p99 %.1f us  (launch + sync, from C)\n", S,
           lat[TICKS / 2], lat[(int)(TICKS * 0.99)]);
    CHECK(savgol_streambank_samples_received(bank) == 100 + TICKS);

    savgol_streambank_destroy(bank);
    savgol_destroy(f);
    hipFree(d_in); hipFree(d_out); hipFree(d_s); hipFree(d_o);
    free(x); free(y); free(yb);
    printf("c_api_demo: OK\n");
    return 0;
}
