#define _GNU_SOURCE
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

/* Latency-critical callers belong on the CPU socket the GPU hangs off: a doorbell or a launch that crosses the socket
 * interconnect first costs microseconds (measured on 2-socket MI355X hosts: 7-8 us vs 12 us per resident-service tick).  The
 * kernel tells which CPUs are local in sysfs; pin this thread to them.  Returns the number of CPUs in the mask (0 = not pinned). */
#define _GNU_SOURCE_PIN
#include <sched.h>
static int pin_to_gpu_numa_node(void)
{
    char bus[64], path[160], list[1024];
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, 0) != hipSuccess) return 0;
    for (char *c = bus; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bus);
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    if (!fgets(list, sizeof list, f)) { fclose(f); return 0; }
    fclose(f);
    cpu_set_t set;
    CPU_ZERO(&set);
    int count = 0;
    for (char *p = list; *p && *p != '\n';) {
        char *end;
        long a = strtol(p, &end, 10), b = a;
        if (end == p) break;
        if (*end == '-') { p = end + 1; b = strtol(p, &end, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, &set); ++count; }
        p = (*end == ',') ? end + 1 : end;
    }
    if (count == 0 || sched_setaffinity(0, sizeof set, &set) != 0) return 0;
    return count;
}

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
    /* ... and bit for bit when the batch call is asked for the reference's order too: a per-call flag, so another thread's
     * calls keep their own summation order (the process-wide savgol_hip_set_option only sets the default of the non-_ex calls) */
    CHECK(savgol_apply_batch_f32_ex(f, d_in, d_out, CH, L, L, L, SAVGOL_BATCH_REFERENCE_SUMMATION, NULL) == 0);
    CHECK(hipMemcpy(yb, d_out, sizeof(float) * L * CH, hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(memcmp(yb + 5 * L, y, sizeof(float) * L) == 0);
    double worst = 0.0;                                          /* smoothing a smooth signal: stays close to it */
    for (size_t i = 100; i < L - 100; ++i) { double d = fabs(yb[i] - sinf(0.001f * (float)i) - 0.05); if (d > worst) worst = d; }
    CHECK(worst < 0.05);
    printf("batch: %zu channels x %zu samples filtered on the GPU; drop-in and device-resident results %.1e apart, identical in reference-order mode\n", CH, L, apart);

    /* ---- 3. 65 536 concurrent streams, one tick per launch ---- */
    printf("pinned to the GPU's NUMA node: %d local CPUs\n", pin_to_gpu_numa_node());
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
    {
        /* the same tick with the completion taken from a stream-written word instead of hipStreamSynchronize (savgol_streambank_push_wait) */
        for (int t = 0; t < 100; ++t) CHECK(savgol_streambank_push_wait(bank, d_s, d_o, NULL) == 1);
        for (int t = 0; t < TICKS; ++t) {
            const double t0 = now_us();
            CHECK(savgol_streambank_push_wait(bank, d_s, d_o, NULL) == 1);
            lat[t] = now_us() - t0;
        }
        qsort(lat, TICKS, sizeof(double), cmp_double);
        printf("stream bank: %zu streams, n=16 m=2 d=1: per-tick wall latency p50 %.1f us  p99 %.1f us  (push_wait: launch + stream-written completion word, from C)\n", S,
               lat[TICKS / 2], lat[(int)(TICKS * 0.99)]);
    }
    {
        /* the same ticks enqueued back to back with no synchronise in between, HIP events around the lot: the device time per tick when the
         * kernel is the slower side, the host's launch rate when it is not (the smaller of the two is not visible from here; rocprofv3's
         * kernel trace has the kernel alone: profiles/README.md) */
        hipEvent_t e0, e1;
        float ms = 0.0f;
        CHECK(hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess);
        CHECK(hipEventRecord(e0, NULL) == hipSuccess);
        for (int t = 0; t < TICKS; ++t) CHECK(savgol_streambank_push(bank, d_s, d_o, NULL) == 1);
        CHECK(hipEventRecord(e1, NULL) == hipSuccess);
        CHECK(hipEventSynchronize(e1) == hipSuccess);
        CHECK(hipEventElapsedTime(&ms, e0, e1) == hipSuccess);
        printf("stream bank: %zu streams, n=16 m=2 d=1: %.2f us per tick back to back (%d launches enqueued from C, no synchronise between them)\n", S,
               (double)ms * 1e3 / TICKS, (int)TICKS);
        hipEventDestroy(e0); hipEventDestroy(e1);
        CHECK(savgol_streambank_samples_received(bank) == 200 + 3 * TICKS);
    }

    /* ---- 3b. the same ticks through the resident service: a doorbell instead of a launch + synchronise per tick.  Noisy
     *          input this time, and every tick's output compared bit for bit with the per-tick kernel on a twin bank ---- */
    {
        enum { ROWS = 64 };
        SavgolStreamBank *twin = savgol_streambank_create(&scfg, S);
        CHECK(twin != NULL);
        float *h_rows = (float *)malloc(sizeof(float) * S * ROWS), *d_rows, *d_o2;
        float *h_a = (float *)malloc(sizeof(float) * S), *h_b = (float *)malloc(sizeof(float) * S);
        unsigned rs = 12345u;
        for (size_t i = 0; i < S * ROWS; ++i) { rs = rs * 1664525u + 1013904223u; h_rows[i] = (float)(rs >> 8) * (1.0f / 8388608.0f) - 1.0f; }
        CHECK(hipMalloc((void **)&d_rows, sizeof(float) * S * ROWS) == hipSuccess);
        CHECK(hipMalloc((void **)&d_o2, sizeof(float) * S) == hipSuccess);
        CHECK(hipMemcpy(d_rows, h_rows, sizeof(float) * S * ROWS, hipMemcpyHostToDevice) == hipSuccess);
        CHECK(savgol_streambank_reset(bank, NULL) == 0);
        CHECK(savgol_hip_synchronize(NULL) == 0);
        CHECK(savgol_streambank_service_start(bank, 2000) == 0);
        for (int t = 0; t < 100; ++t) {                          /* warm-up incl. the filling phase; compare every tick */
            const int r1 = savgol_streambank_service_tick(bank, d_rows + (size_t)(t % ROWS) * S, d_o);
            const int r2 = savgol_streambank_push(twin, d_rows + (size_t)(t % ROWS) * S, d_o2, NULL);
            CHECK(r1 == r2 && r1 >= 0);
            if (r1 == 1) {
                CHECK(hipMemcpy(h_a, d_o, sizeof(float) * S, hipMemcpyDeviceToHost) == hipSuccess);
                CHECK(hipMemcpy(h_b, d_o2, sizeof(float) * S, hipMemcpyDeviceToHost) == hipSuccess);
                CHECK(memcmp(h_a, h_b, sizeof(float) * S) == 0);
            }
        }
        for (int t = 0; t < TICKS; ++t) {
            const double t0 = now_us();
            CHECK(savgol_streambank_service_tick(bank, d_rows + (size_t)(t % ROWS) * S, d_o) == 1);
            lat[t] = now_us() - t0;
        }
        CHECK(savgol_streambank_service_stop(bank) == 0);
        qsort(lat, TICKS, sizeof(double), cmp_double);
        printf("stream bank: %zu streams, n=16 m=2 d=1: per-tick wall latency p50 %.1f us  p99 %.1f us  (resident service: doorbell + completion array, from C; outputs bit-identical to the per-tick kernel)\n",
               S, lat[TICKS / 2], lat[(int)(TICKS * 0.99)]);
        CHECK(savgol_streambank_samples_received(bank) == 100 + TICKS);
        savgol_streambank_destroy(twin);
        hipFree(d_rows); hipFree(d_o2); free(h_rows); free(h_a); free(h_b);
    }

    /* ---- 3c. many ticks per launch: 1024 ticks of all the streams in one call.  The bank created above keeps the reference's
     *          summation order (bit-identical to savgol_stream_push); a bank created with SAVGOL_STREAMBANK_FMA runs fused
     *          multiply-adds instead -- faster, and within fp32 rounding of the first ---- */
    {
        enum { BT = 1024 };
        SavgolStreamBank *fast = savgol_streambank_create_ex(&scfg, S, SAVGOL_STREAMBANK_FMA);
        CHECK(fast != NULL);
        float *h_blk = (float *)malloc(sizeof(float) * S * BT), *h_y1 = (float *)malloc(sizeof(float) * S), *h_y2 = (float *)malloc(sizeof(float) * S);
        float *d_blk, *d_y1, *d_y2;
        unsigned rs = 777u;
        for (size_t i = 0; i < S * BT; ++i) { rs = rs * 1664525u + 1013904223u; h_blk[i] = (float)(rs >> 8) * (1.0f / 8388608.0f) - 1.0f; }
        CHECK(hipMalloc((void **)&d_blk, sizeof(float) * S * BT) == hipSuccess);
        CHECK(hipMalloc((void **)&d_y1, sizeof(float) * S * BT) == hipSuccess);
        CHECK(hipMalloc((void **)&d_y2, sizeof(float) * S * BT) == hipSuccess);
        CHECK(hipMemcpy(d_blk, h_blk, sizeof(float) * S * BT, hipMemcpyHostToDevice) == hipSuccess);
        CHECK(savgol_streambank_reset(bank, NULL) == 0);
        double ms[2] = {0.0, 0.0};
        for (int rep = 0; rep < 3; ++rep) {                      /* windows fill during the first call; time the last */
            SavgolStreamBank *banks[2] = {bank, fast};
            float *outs[2] = {d_y1, d_y2};
            for (int b = 0; b < 2; ++b) {
                CHECK(savgol_hip_synchronize(NULL) == 0);
                const double t0 = now_us();
                CHECK(savgol_streambank_push_block(banks[b], d_blk, BT, outs[b], NULL) == (rep == 0 ? BT - 32 : BT));
                CHECK(savgol_hip_synchronize(NULL) == 0);
                ms[b] = (now_us() - t0) * 1e-3;
            }
        }
        CHECK(hipMemcpy(h_y1, d_y1 + (size_t)(BT - 1) * S, sizeof(float) * S, hipMemcpyDeviceToHost) == hipSuccess);
        CHECK(hipMemcpy(h_y2, d_y2 + (size_t)(BT - 1) * S, sizeof(float) * S, hipMemcpyDeviceToHost) == hipSuccess);
        double top = 0.0, gap = 0.0;
        for (size_t i = 0; i < S; ++i) {
            if (fabs(h_y1[i]) > top) top = fabs(h_y1[i]);
            if (fabs((double)h_y1[i] - (double)h_y2[i]) > gap) gap = fabs((double)h_y1[i] - (double)h_y2[i]);
        }
        CHECK(gap <= 4e-6 * top);
        printf("stream bank: %d ticks x %zu streams per call: %.3f ms in the reference's order, %.3f ms with fused multiply-adds (%.0f / %.0f Gsamples/s), last tick %.1e apart (relative)\n",
               BT, S, ms[0], ms[1], 1e-6 * S * BT / ms[0], 1e-6 * S * BT / ms[1], gap / top);
        savgol_streambank_destroy(fast);
        hipFree(d_blk); hipFree(d_y1); hipFree(d_y2); free(h_blk); free(h_y1); free(h_y2);
    }

    savgol_streambank_destroy(bank);
    savgol_destroy(f);
    hipFree(d_in); hipFree(d_out); hipFree(d_s); hipFree(d_o);
    free(x); free(y); free(yb);
    printf("c_api_demo: OK\n");
    return 0;
}
