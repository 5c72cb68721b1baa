/* cpu_stream_bench.c -- times the REFERENCE's savgol_stream_push loop (1 stream, 1 thread) on the host.
 * TEST/BENCH INFRASTRUCTURE: linked against oracle/_ref/libsavgol_ref.so (the compiled, unmodified reference);
 * built by `make -C oracle ref` into oracle/_ref/cpu_stream_bench; used by tools/bench_paths.py as cpu_baseline.
 *   usage: cpu_stream_bench <half_window> <poly_order> <derivative> <time_step> <samples>      -> prints Msamples/s */
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "savgol_stream.h"      /* the reference's header (include path set by the Makefile) */

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    SavgolConfig cfg = { .half_window = (uint8_t)atoi(argv[1]), .poly_order = (uint8_t)atoi(argv[2]),
                         .derivative = (uint8_t)atoi(argv[3]), .time_step = (float)atof(argv[4]),
                         .boundary = SAVGOL_BOUNDARY_POLYNOMIAL };
    const long n = atol(argv[5]);
    SavgolStream *s = savgol_stream_create(&cfg);
    if (!s) return 1;
    float *x = malloc(sizeof(float) * n);
    unsigned z = 12345u;
    for (long i = 0; i < n; ++i) { z = z * 1664525u + 1013904223u; x[i] = (float)(z >> 8) * (1.0f / 16777216.0f) - 0.5f; }
    double best = 1e30;
    volatile float sink = 0.0f;
    for (int rep = 0; rep < 5; ++rep) {
        savgol_stream_reset(s);
        struct timespec a, b;
        bool ok;
        float acc = 0.0f;
        clock_gettime(CLOCK_MONOTONIC, &a);
        for (long i = 0; i < n; ++i) acc += savgol_stream_push(s, x[i], &ok);
        clock_gettime(CLOCK_MONOTONIC, &b);
        sink += acc;
        const double el = (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
        if (el < best) best = el;
    }
    printf("%.3f\n", n / best / 1e6);
    savgol_stream_destroy(s);
    free(x);
    return 0;
}
