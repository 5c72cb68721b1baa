// sg_2d_rowband.hip -- a frame split into horizontal row bands, one band per GPU (the one exchange step on the hot
// path), through the C ABI.
//
// Reference semantics to reproduce at an ARTIFICIAL band edge: none -- the reference filters whole frames (savgol2d_apply,
// src/savgol2d.c:398-456; its boundary modes, :417-453, apply at the frame's real edges only), so a band's output rows must
// be exactly the rows the whole-frame call would have produced.  Output row r reads input rows r-ny .. r+ny: a band needs
// the ny rows just above and just below it from its neighbours ("halo rows"), and nothing else.
//
//   savgol2d_rowband_plan        which rows a rank owns and how many halo rows it needs from either side
//   savgol2d_apply_rowband_f32   filters one band given its halo rows (device buffers the caller filled -- a device-to-device
//                                copy on one GPU, ncclSend/ncclRecv across GPUs: sg_rowband_rccl.cpp)
//
// How (as rowband.py's apply_overlapped did in Python since round 2): (1) the band is filtered as if it were a frame -- every
// output row except the ny next to an artificial edge is already final, and at a REAL frame edge (halo pointer NULL) the
// boundary mode applies as in the reference; this launch depends on no halo, so a caller can enqueue it while the exchange is
// still in flight on another stream.  (2) Each artificial edge is redone on a 3 ny-row strip (halo rows + the band's first /
// last 2 ny rows) filtered as a frame: its middle ny output rows see only real data and replace the tainted ones.  The strips
// are 3 ny / band_rows of the work (0.5 % for 4096-row bands at ny = 7).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "sg_2d.hpp"
#include "sg_runtime.hpp"

namespace sg {

// rows x cols floats per image, image k at base + k * pitch
__global__ __launch_bounds__(256) void sg2d_copy_rows_kernel(float *__restrict__ dst, int dst_stride, long long dst_pitch,
                                                             const float *__restrict__ src, int src_stride, long long src_pitch,
                                                             int rows, int cols)
{
    const long long img = blockIdx.z;
    const int r = blockIdx.y;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < cols; c += gridDim.x * 256)
        dst[img * dst_pitch + (long long)r * dst_stride + c] = src[img * src_pitch + (long long)r * src_stride + c];
    (void)rows;
}

static bool copy_rows(float *dst, int dst_stride, long long dst_pitch, const float *src, int src_stride, long long src_pitch, int rows,
                      int cols, size_t images, hipStream_t st)
{
    if (rows <= 0 || cols <= 0) return true;
    unsigned gx = (unsigned)((cols + 255) / 256);
    if (gx > 64) gx = 64;
    for (size_t i0 = 0; i0 < images; i0 += 65535) {
        const size_t ni = images - i0 < 65535 ? images - i0 : 65535;
        hipLaunchKernelGGL(sg2d_copy_rows_kernel, dim3(gx, (unsigned)rows, (unsigned)ni), dim3(256), 0, st, dst + (long long)i0 * dst_pitch, dst_stride,
                           dst_pitch, src + (long long)i0 * src_pitch, src_stride, src_pitch, rows, cols);
    }
    return hip_ok(hipGetLastError(), "row copy launch");
}

}  // namespace sg

extern "C" {

int savgol2d_rowband_plan(int rows, int half_win_y, int rank, int world_size, int *row_lo, int *row_hi, int *halo_up, int *halo_down)
{
    size_t lo = 0, hi = 0;
    if (rows <= 0 || half_win_y < 1 || !row_lo || !row_hi || savgol_hip_shard_range((size_t)rows, world_size, rank, &lo, &hi) != 0) {
        sg_set_error("savgol2d_rowband_plan: bad arguments");
        return -1;
    }
    // every band must hold the 2 ny rows an artificial edge is rebuilt from (and the ny rows its neighbour needs)
    if (world_size > 1 && rows / world_size < 2 * half_win_y) {
        sg_set_error("savgol2d_rowband_plan: %d rows over %d ranks leaves bands thinner than 2 x half_window_y = %d: use fewer ranks", rows,
                     world_size, 2 * half_win_y);
        return -1;
    }
    *row_lo = (int)lo; *row_hi = (int)hi;
    if (halo_up) *halo_up = rank > 0 ? half_win_y : 0;
    if (halo_down) *halo_down = rank + 1 < world_size ? half_win_y : 0;
    return 0;
}

static int rowband_check(const char *who, const Savgol2DFilter *filter, const float *d_band, float *d_out, int band_rows, int cols, const float *d_halo_up,
                         const float *d_halo_down, int halo_stride, size_t halo_image_pitch, size_t images)
{
    if (!filter || !d_band || !d_out) { sg_set_error("%s: NULL pointer", who); return -1; }
    const int nx = filter->config.half_window_x, ny = filter->config.half_window_y;
    if (nx < 1 || ny < 1 || nx > SAVGOL2D_MAX_HALF_WINDOW || ny > SAVGOL2D_MAX_HALF_WINDOW) { sg_set_error("%s: filter struct is not a valid Savgol2DFilter", who); return -1; }
    const bool any_halo = d_halo_up || d_halo_down;
    if (any_halo && band_rows < 2 * ny) {
        sg_set_error("%s: a band with a neighbour needs >= 2 x half_window_y = %d rows (got %d): its artificial edges are rebuilt from them", who,
                     2 * ny, band_rows);
        return -1;
    }
    if (any_halo && (halo_stride < cols || (images > 1 && halo_image_pitch < (size_t)ny * (size_t)halo_stride))) { sg_set_error("%s: bad halo geometry", who); return -1; }
    return 0;
}

// step (2) alone: the ny output rows next to each artificial edge, from the halo rows and the band's own 2 ny rows next to that edge
int savgol2d_apply_rowband_edges_f32(const Savgol2DFilter *filter, const float *d_band, int band_rows, int cols, int in_stride,
                                     size_t in_image_pitch, const float *d_halo_up, const float *d_halo_down, int halo_stride,
                                     size_t halo_image_pitch, float *d_out, int out_stride, size_t out_image_pitch, size_t images,
                                     Savgol2DBoundary boundary, int method, void *stream)
{
    const char *who = "savgol2d_apply_rowband_edges_f32";
    if (rowband_check(who, filter, d_band, d_out, band_rows, cols, d_halo_up, d_halo_down, halo_stride, halo_image_pitch, images) != 0) return -1;
    if ((!d_halo_up && !d_halo_down) || images == 0) return 0;
    const int nx = filter->config.half_window_x, ny = filter->config.half_window_y;
    if (cols - 2 * nx <= 0 && boundary == SAVGOL2D_BOUNDARY_VALID) { sg_set_error("%s: image smaller than the window", who); return -1; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool valid = boundary == SAVGOL2D_BOUNDARY_VALID;
    // [halo | first 2 ny rows] and [last 2 ny rows | halo], each 3 ny rows, filtered as frames
    const int srows = 3 * ny, sstride = (cols + 3) & ~3;
    const size_t simg = (size_t)srows * sstride;
    const int nstrips = (d_halo_up ? 1 : 0) + (d_halo_down ? 1 : 0);
    float *scratch = nullptr;
    sg::DeviceCtx *ctx = sg::ctx_get();
    if (!ctx) return -1;
    scratch = static_cast<float *>(sg::scratch_alloc(ctx, sizeof(float) * 2 * simg * images * nstrips, st, "scratch (row-band strips)"));
    if (!scratch) return -1;
    int rc = 0;
    int k = 0;
    for (int side = 0; side < 2 && rc == 0; ++side) {
        const float *halo = side == 0 ? d_halo_up : d_halo_down;
        if (!halo) continue;
        float *sin = scratch + (size_t)(2 * k) * simg * images, *sout = sin + simg * images;
        ++k;
        const float *own = side == 0 ? d_band : d_band + (size_t)(band_rows - 2 * ny) * in_stride;       // the band's 2 ny rows next to this edge
        bool ok;
        if (side == 0)
            ok = sg::copy_rows(sin, sstride, (long long)simg, halo, halo_stride, (long long)halo_image_pitch, ny, cols, images, st) &&
                 sg::copy_rows(sin + (size_t)ny * sstride, sstride, (long long)simg, own, in_stride, (long long)in_image_pitch, 2 * ny, cols, images, st);
        else
            ok = sg::copy_rows(sin, sstride, (long long)simg, own, in_stride, (long long)in_image_pitch, 2 * ny, cols, images, st) &&
                 sg::copy_rows(sin + (size_t)(2 * ny) * sstride, sstride, (long long)simg, halo, halo_stride, (long long)halo_image_pitch, ny, cols, images, st);
        if (!ok) { rc = -1; break; }
        if (savgol2d_apply_batch_f32(filter, sin, srows, cols, sstride, simg, sout, sstride, simg, images, boundary, method, stream) != 0) { rc = -1; break; }
        // the strip's middle ny output rows are band rows [0, ny) / [band_rows - ny, band_rows); VALID wrote columns [nx, cols - nx) only
        const int c0 = valid ? nx : 0, nc = valid ? cols - 2 * nx : cols;
        float *dst = (side == 0 ? d_out : d_out + (size_t)(band_rows - ny) * out_stride) + c0;
        if (!sg::copy_rows(dst, out_stride, (long long)out_image_pitch, sout + (size_t)ny * sstride + c0, sstride, (long long)simg, ny, nc, images, st)) rc = -1;
    }
    if (!sg::scratch_free(scratch, st, "scratch free (row-band strips)")) rc = -1;
    return rc;
}

int savgol2d_apply_rowband_f32(const Savgol2DFilter *filter, const float *d_band, int band_rows, int cols, int in_stride,
                               size_t in_image_pitch, const float *d_halo_up, const float *d_halo_down, int halo_stride,
                               size_t halo_image_pitch, float *d_out, int out_stride, size_t out_image_pitch, size_t images,
                               Savgol2DBoundary boundary, int method, void *stream)
{
    const char *who = "savgol2d_apply_rowband_f32";
    if (rowband_check(who, filter, d_band, d_out, band_rows, cols, d_halo_up, d_halo_down, halo_stride, halo_image_pitch, images) != 0) return -1;
    const int nx = filter->config.half_window_x, ny = filter->config.half_window_y;
    const bool valid = boundary == SAVGOL2D_BOUNDARY_VALID;
    // (1) the band as a frame.  VALID on a band thinner than the window writes nothing here; its rows all come from the strips.
    if (!(valid && (band_rows - 2 * ny <= 0))) {
        if (savgol2d_apply_batch_f32(filter, d_band, band_rows, cols, in_stride, in_image_pitch, d_out, out_stride, out_image_pitch, images,
                                     boundary, method, stream) != 0) return -1;
    } else if (cols - 2 * nx <= 0) { sg_set_error("%s: image smaller than the window", who); return -1; }
    // (2) the artificial edges
    return savgol2d_apply_rowband_edges_f32(filter, d_band, band_rows, cols, in_stride, in_image_pitch, d_halo_up, d_halo_down, halo_stride,
                                            halo_image_pitch, d_out, out_stride, out_image_pitch, images, boundary, method, stream);
}

}  // extern "C"
