/*
 * rowband_demo.c -- the 2-D row-band split through the C ABI, on ONE GPU.
 *
 *   make -C savitzky-golay-filter_amd     ->  savitzky-golay-filter_amd/lib/rowband_demo
 *
 * A stack of frames is filtered (a) whole, by savgol2d_apply_batch_f32, and (b) as `world` row bands, each by
 * savgol2d_apply_rowband_f32 with the halo rows its neighbours would have sent -- copied device to device here, where a
 * multi-GPU host would run savgol2d_rowband_exchange_rccl (libsavgol_hip_rccl.so); with three bands through the split form of a
 * multi-GPU step: savgol2d_apply_batch_f32 on the band, savgol2d_apply_rowband_edges_streams_f32 on a second stream.  The stitched bands must equal the whole
 * frames: bit for bit with method 1 (and method 2 where the kernel is not the additive rolling form), for VALID, CONSTANT and
 * REFLECT, bands of unequal height included.  Exit code 0 = all checks passed.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "savgol2d.h"
#include "savgol_hip.h"

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "FAILED line %d: %s (%s)\n", __LINE__, #x, savgol_hip_last_error()); return 1; } } while (0)

static unsigned long long rng_state = 0x5A17601AULL;
static float rnd(void)
{
    rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
    return (float)((rng_state >> 40) & 0xffff) / 32768.0f - 1.0f;
}

int main(void)
{
    CHECK(savgol_hip_device_count() > 0);
    const int images = 3, rows = 173, cols = 260, stride = 264;
    const size_t frame = (size_t)rows * stride;
    float *h = (float *)malloc(sizeof(float) * frame * images), *whole = (float *)malloc(sizeof(float) * frame * images),
          *parts = (float *)malloc(sizeof(float) * frame * images);
    for (size_t i = 0; i < frame * images; ++i) h[i] = rnd();
    float *d_in, *d_whole, *d_parts, *d_halo;
    CHECK(hipMalloc((void **)&d_in, sizeof(float) * frame * images) == hipSuccess);
    CHECK(hipMalloc((void **)&d_whole, sizeof(float) * frame * images) == hipSuccess);
    CHECK(hipMalloc((void **)&d_parts, sizeof(float) * frame * images) == hipSuccess);
    CHECK(hipMemcpy(d_in, h, sizeof(float) * frame * images, hipMemcpyHostToDevice) == hipSuccess);

    hipStream_t side;
    CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking) == hipSuccess);
    const int configs[][5] = {{7, 7, 3, 0, 0}, {3, 3, 2, 1, 0}, {5, 5, 4, 0, 0}, {4, 6, 3, 0, 0}};     /* nx, ny, order, dx, dy */
    int checks = 0;
    for (unsigned c = 0; c < sizeof configs / sizeof configs[0]; ++c) {
        Savgol2DConfig cfg = {(uint8_t)configs[c][0], (uint8_t)configs[c][1], (uint8_t)configs[c][2], (uint8_t)configs[c][3], (uint8_t)configs[c][4], 1.0f, 1.0f};
        Savgol2DFilter *f = savgol2d_create(&cfg);
        CHECK(f != NULL);
        const int ny = cfg.half_window_y, square = cfg.half_window_x == cfg.half_window_y;
        CHECK(hipMalloc((void **)&d_halo, sizeof(float) * 2 * (size_t)images * ny * cols) == hipSuccess);
        for (int world = 2; world <= 4; ++world)
            for (int boundary = 0; boundary < 3; ++boundary)
                for (int method = 1; method <= (square ? 2 : 1); ++method) {
                    CHECK(hipMemset(d_whole, 0x7f, sizeof(float) * frame * images) == hipSuccess);       /* same fill: untouched pixels compare equal */
                    CHECK(hipMemset(d_parts, 0x7f, sizeof(float) * frame * images) == hipSuccess);
                    CHECK(savgol2d_apply_batch_f32(f, d_in, rows, cols, stride, frame, d_whole, stride, frame, images, (Savgol2DBoundary)boundary, method, NULL) == 0);
                    for (int rank = 0; rank < world; ++rank) {
                        int lo, hi, up, down;
                        CHECK(savgol2d_rowband_plan(rows, ny, rank, world, &lo, &hi, &up, &down) == 0);
                        CHECK(up == (rank > 0 ? ny : 0) && down == (rank + 1 < world ? ny : 0));
                        /* what the neighbours would send: rows [lo - ny, lo) and [hi, hi + ny) of every frame, packed [image][ny][cols] */
                        float *halo_up = d_halo, *halo_down = d_halo + (size_t)images * ny * cols;
                        for (int k = 0; k < images; ++k) {
                            if (up) CHECK(hipMemcpy2D(halo_up + (size_t)k * ny * cols, sizeof(float) * cols, d_in + k * frame + (size_t)(lo - ny) * stride,
                                                      sizeof(float) * stride, sizeof(float) * cols, ny, hipMemcpyDeviceToDevice) == hipSuccess);
                            if (down) CHECK(hipMemcpy2D(halo_down + (size_t)k * ny * cols, sizeof(float) * cols, d_in + k * frame + (size_t)hi * stride,
                                                        sizeof(float) * stride, sizeof(float) * cols, ny, hipMemcpyDeviceToDevice) == hipSuccess);
                        }
                        if (world != 3) {
                            CHECK(savgol2d_apply_rowband_f32(f, d_in + (size_t)lo * stride, hi - lo, cols, stride, frame, up ? halo_up : NULL, down ? halo_down : NULL,
                                                             cols, (size_t)ny * cols, d_parts + (size_t)lo * stride, stride, frame, images,
                                                             (Savgol2DBoundary)boundary, method, NULL) == 0);
                        } else {
                            /* the split form a multi-GPU step uses: the band on the compute stream (it reads no halo); the edge strips gathered and
                             * filtered on the stream the halos arrived on, beside the band; their finished rows copied in behind it */
                            CHECK(hipStreamSynchronize(NULL) == hipSuccess);      /* the device-to-device halo copies above are the "exchange": done before `side` reads them */
                            if (!(boundary == SAVGOL2D_BOUNDARY_VALID && hi - lo - 2 * ny <= 0))
                                CHECK(savgol2d_apply_batch_f32(f, d_in + (size_t)lo * stride, hi - lo, cols, stride, frame, d_parts + (size_t)lo * stride, stride, frame,
                                                               images, (Savgol2DBoundary)boundary, method, NULL) == 0);
                            CHECK(savgol2d_apply_rowband_edges_streams_f32(f, d_in + (size_t)lo * stride, hi - lo, cols, stride, frame, up ? halo_up : NULL,
                                                                           down ? halo_down : NULL, cols, (size_t)ny * cols, d_parts + (size_t)lo * stride, stride,
                                                                           frame, images, (Savgol2DBoundary)boundary, method, side, NULL) == 0);
                            CHECK(hipStreamSynchronize(side) == hipSuccess);      /* the next rank's hipMemcpy2D re-uses d_halo */
                        }
                    }
                    CHECK(hipDeviceSynchronize() == hipSuccess);
                    CHECK(hipMemcpy(whole, d_whole, sizeof(float) * frame * images, hipMemcpyDeviceToHost) == hipSuccess);
                    CHECK(hipMemcpy(parts, d_parts, sizeof(float) * frame * images, hipMemcpyDeviceToHost) == hipSuccess);
                    /* the additive smoothing kernels (order <= 3, no derivative) run rolling column sums in method 2: rounding, not bits */
                    const int additive = method == 2 && cfg.poly_order <= 3 && cfg.deriv_x == 0 && cfg.deriv_y == 0;
                    if (!additive) {
                        if (memcmp(whole, parts, sizeof(float) * frame * images) != 0) {
                            size_t bad = 0, first = (size_t)-1;
                            for (size_t i = 0; i < frame * images; ++i)
                                if (memcmp(&whole[i], &parts[i], 4) != 0) { if (first == (size_t)-1) first = i; ++bad; }
                            fprintf(stderr, "FAILED: config %u world %d boundary %d method %d: stitched bands differ from the whole frame in %zu values; first at image %zu row %zu column %zu: %.9g vs %.9g\n",
                                    c, world, boundary, method, bad, first / frame, (first % frame) / stride, first % stride, whole[first], parts[first]);
                            return 1;
                        }
                    } else {
                        double worst = 0.0, vmax = 1.0;              /* the rolling column sums scale with the INPUT, here uniform in [-1, 1] */
                        for (size_t i = 0; i < frame * images; ++i) {
                            unsigned a, b;
                            memcpy(&a, &whole[i], 4); memcpy(&b, &parts[i], 4);
                            if ((a == 0x7f7f7f7fu) != (b == 0x7f7f7f7fu)) { fprintf(stderr, "FAILED: written regions differ (config %u)\n", c); return 1; }
                            if (a == 0x7f7f7f7fu) continue;
                            if (fabs((double)whole[i] - parts[i]) > worst) worst = fabs((double)whole[i] - parts[i]);
                        }
                        if (!(worst <= 2.5e-7 * vmax)) { fprintf(stderr, "FAILED: config %u world %d boundary %d: %g of %g\n", c, world, boundary, worst, vmax); return 1; }
                    }
                    ++checks;
                }
        /* bands thinner than 2 ny rows are refused, not silently wrong */
        int lo, hi;
        CHECK(savgol2d_rowband_plan(4 * ny - 1, ny, 0, 2, &lo, &hi, NULL, NULL) == -1);
        CHECK(savgol2d_apply_rowband_f32(f, d_in, 2 * ny - 1, cols, stride, frame, d_halo, NULL, cols, (size_t)ny * cols, d_parts, stride, frame, images,
                                         SAVGOL2D_BOUNDARY_CONSTANT, 1, NULL) == -1);
        CHECK(hipFree(d_halo) == hipSuccess);
        savgol2d_destroy(f);
    }
    printf("rowband_demo: OK (%d band-split / whole-frame comparisons)\n", checks);
    return 0;
}
