#define _GNU_SOURCE
/*
 * time_demo360.c -- ONE program, ONE clock for the small-signal claim: the reference's own demo shape
 * (test/iterative/test_savgol_main.c:136-155: 360 points, half_window 6, poly_order 3, savgol_apply called 10 000 times back to back)
 * timed with clock_gettime on every library given -- the compiled reference (oracle/_ref/libsavgol_ref.so) and the GPU drop-in
 * (lib/libsavgol_hip.so) -- in the same process, same input, same loop.  Round 3's documents quoted three different CPU figures for
 * this (36-42, 74.8 and 250.7 Msamples/s: a demo binary's clock(), ctypes in Python, and `-O2` in another container); bench.py and
 * the READMEs now quote this program's output only.
 *
 *   gcc -O2 -o lib/time_demo360 tools/time_demo360.c -ldl -lm
 *   time_demo360 oracle/_ref/libsavgol_ref.so savitzky-golay-filter_amd/lib/libsavgol_hip.so
 *
 * Per library: Msamples/s of the loop, microseconds per call (median of 10 blocks of 1000 calls), and -- for a library that exports
 * savgol_hip_synchronize -- the same call followed by a device-wide synchronise (the mixed workload ADVICE r03 asked for: what a
 * caller pays who synchronises right after a short call while the small-call service's workgroup is still resident).
 * The first output sample is printed so that the two libraries can be seen to compute the same thing.
 */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct { uint8_t half_window, poly_order, derivative; float time_step; int boundary; } Config;   /* savgolFilter.h:92-98 */
typedef void *(*create_fn)(const Config *);
typedef void (*destroy_fn)(void *);
typedef int (*apply_fn)(const void *, const float *, float *, size_t);
typedef int (*sync_fn)(void *);
typedef int (*devsync_fn)(void);

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
    enum { N = 360, BLOCKS = 10, PER = 1000 };
    float in[N], out[N];
    for (int i = 0; i < N; ++i) in[i] = 10.0f + sinf(0.05f * i) + 0.3f * sinf(0.9f * i + 1.0f);
    printf("{");
    for (int a = 1; a < argc; ++a) {
        void *h = dlopen(argv[a], RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "%s\n", dlerror()); return 1; }
        create_fn create = (create_fn)dlsym(h, "savgol_create");
        destroy_fn destroy = (destroy_fn)dlsym(h, "savgol_destroy");
        apply_fn apply = (apply_fn)dlsym(h, "savgol_apply");
        sync_fn sync = (sync_fn)dlsym(h, "savgol_hip_synchronize");
        if (!create || !destroy || !apply) { fprintf(stderr, "%s: missing symbols\n", argv[a]); return 1; }
        Config cfg = {6, 3, 0, 1.0f, 0};
        void *f = create(&cfg);
        if (!f) return 1;
        for (int i = 0; i < 200; ++i) if (apply(f, in, out, N) != 0) { fprintf(stderr, "%s: savgol_apply failed\n", argv[a]); return 1; }
        double us[BLOCKS];
        const double t_all = now_us();
        for (int b = 0; b < BLOCKS; ++b) {
            const double t0 = now_us();
            for (int i = 0; i < PER; ++i) (void)apply(f, in, out, N);
            us[b] = (now_us() - t0) / PER;
        }
        const double total = now_us() - t_all;
        qsort(us, BLOCKS, sizeof(double), cmp);
        printf("%s\"%s\": {\"Msamples_per_s\": %.2f, \"us_per_call_median\": %.3f, \"us_per_call_min\": %.3f, \"out0\": %.6f", a > 1 ? ", " : "", argv[a],
               (double)N * BLOCKS * PER / total, us[BLOCKS / 2], us[0], out[0]);
        if (sync) {
            /* the mixed workload: short call, then a device-wide wait (hipDeviceSynchronize through the HIP runtime the library loaded) */
            void *hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_NOLOAD);
            devsync_fn devsync = hip ? (devsync_fn)dlsym(hip, "hipDeviceSynchronize") : NULL;
            if (devsync) {
                double m[200];
                for (int i = 0; i < 200; ++i) {
                    const double t0 = now_us();
                    (void)apply(f, in, out, N);
                    (void)devsync();
                    m[i] = now_us() - t0;
                }
                qsort(m, 200, sizeof(double), cmp);
                printf(", \"call_then_device_synchronize_us_median\": %.2f, \"call_then_device_synchronize_us_p99\": %.2f", m[100], m[197]);
            }
        }
        printf("}");
        destroy(f);
    }
    printf("}\n");
    return 0;
}
