#define _GNU_SOURCE
/*
 * time_batch_host.c -- host microseconds per savgol_apply_batch_f32 call.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -o savitzky-golay-filter_amd/lib/time_batch_host \
 *       tools/time_batch_host.c -L/opt/rocm/lib -lamdhip64 -ldl -Wl,-rpath,/opt/rocm/lib
 *   time_batch_host libA.so [libB.so ...]
 *
 * Every library is dlopen()ed in turn (two builds can be compared in one process).  For each (half window, boundary
 * mode): 3000 calls on a tiny job (1 channel x 4096 samples, so the GPU drains the queue faster than the host fills
 * it) timed on the host WITHOUT synchronising -- that is the cost of validating, looking the filter's tables up and
 * enqueueing -- then the same with a synchronise per call on BASELINE config 1's shape (1 x 10^6 samples, end to end).
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <hip/hip_runtime_api.h>

#include "savgolFilter.h"

typedef SavgolFilter *(*create_fn)(const SavgolConfig *);
typedef void (*destroy_fn)(SavgolFilter *);
typedef int (*batch_fn)(const SavgolFilter *, const float *, float *, size_t, size_t, size_t, size_t, void *);

static double now_us(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

static int cmp(const void *a, const void *b) { const double x = *(const double *)a, y = *(const double *)b; return x < y ? -1 : x > y; }

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s lib.so [lib.so ...]\n", argv[0]); return 2; }
    const size_t big = 1000000, small = 4096;
    float *x, *y;
    if (hipMalloc((void **)&x, big * 4) != hipSuccess || hipMalloc((void **)&y, big * 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    (void)hipMemset(x, 0, big * 4);
    for (int a = 1; a < argc; ++a) {
        void *h = dlopen(argv[a], RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "%s\n", dlerror()); return 1; }
        create_fn create = (create_fn)dlsym(h, "savgol_create");
        destroy_fn destroy = (destroy_fn)dlsym(h, "savgol_destroy");
        batch_fn batch = (batch_fn)dlsym(h, "savgol_apply_batch_f32");
        if (!create || !destroy || !batch) { fprintf(stderr, "%s: missing symbols\n", argv[a]); return 1; }
        printf("%s\n", argv[a]);
        const int cases[][3] = {{32, 4, 0}, {32, 4, 1}, {5, 3, 0}, {5, 3, 3}, {16, 2, 0}};
        for (unsigned c = 0; c < sizeof cases / sizeof cases[0]; ++c) {
            SavgolConfig cfg = {(uint8_t)cases[c][0], (uint8_t)cases[c][1], 0, 1.0f, (SavgolBoundaryMode)cases[c][2]};
            SavgolFilter *f = create(&cfg);
            if (!f) return 1;
            for (int i = 0; i < 50; ++i) if (batch(f, x, y, 1, small, small, small, NULL) != 0) { fprintf(stderr, "batch call failed\n"); return 1; }
            (void)hipDeviceSynchronize();
            enum { REPS = 3000 };
            static double t[REPS];
            for (int i = 0; i < REPS; ++i) {
                const double t0 = now_us();
                batch(f, x, y, 1, small, small, small, NULL);
                t[i] = now_us() - t0;
                if ((i & 255) == 255) (void)hipDeviceSynchronize();          /* never let the queue fill up */
            }
            (void)hipDeviceSynchronize();
            qsort(t, REPS, sizeof t[0], cmp);
            /* end to end on config 1's shape: launch + synchronise */
            double e[200];
            for (int i = 0; i < 200; ++i) {
                const double t0 = now_us();
                batch(f, x, y, 1, big, big, big, NULL);
                (void)hipDeviceSynchronize();
                e[i] = now_us() - t0;
            }
            qsort(e, 200, sizeof e[0], cmp);
            printf("  n=%2d mode=%d: enqueue only  p50 %6.2f us  p10 %6.2f  p90 %6.2f   |  1 x 1e6 samples, call + synchronise  p50 %6.2f us\n",
                   cases[c][0], cases[c][2], t[REPS / 2], t[REPS / 10], t[REPS * 9 / 10], e[100]);
            destroy(f);
        }
        /* the launch itself, for scale: an empty-ish HIP call pair on the same stream */
        {
            enum { REPS = 3000 };
            static double t[REPS];
            for (int i = 0; i < REPS; ++i) {
                const double t0 = now_us();
                (void)hipMemsetAsync(y, 0, 64, NULL);
                t[i] = now_us() - t0;
                if ((i & 255) == 255) (void)hipDeviceSynchronize();
            }
            (void)hipDeviceSynchronize();
            qsort(t, REPS, sizeof t[0], cmp);
            printf("  for scale: hipMemsetAsync(64 B) enqueue p50 %6.2f us\n", t[REPS / 2]);
        }
    }
    return 0;
}
