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
// are 3 ny / band_rows of the work (0.5 % for 4096-row bands at ny = 7); with savgol2d_apply_rowband_edges_streams_f32 they are gathered and
// filtered on the exchange's stream, beside the band launch, and only the copy of the finished rows waits for the band (round 6).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "sg_2d.hpp"
#include "sg_runtime.hpp"

namespace sg {

// The two edge strips of every image as one batch of 3 ny-row frames: frame k * images + i = strip k of image i (k = 0: the first strip the band has --
// [halo above | the band's first 2 ny rows], then [the band's last 2 ny rows | halo below]).
struct StripGeom {
    const float *band;                 // image i at band + i * band_pitch
    long long band_pitch;
    int band_stride, band_rows;
    const float *halo[2];              // above / below (nullptr: a real frame edge, no strip); image i at halo + i * halo_pitch
    long long halo_pitch;
    int halo_stride;
    int ny, cols;
    unsigned images;
    int side_of[2];                    // strip k is side_of[k] (0 = above, 1 = below)
};

// gather: strip row r of frame f <- the halo row or band row it stands for
__global__ __launch_bounds__(256) void sg2d_strip_gather_kernel(float *__restrict__ sin, int sstride, long long simg, const StripGeom g, unsigned f0)
{
    const unsigned f = f0 + blockIdx.z, k = f / g.images, i = f - k * g.images;
    const int r = blockIdx.y, side = g.side_of[k], ny = g.ny;
    const float *src;
    if (side == 0) src = r < ny ? g.halo[0] + (long long)i * g.halo_pitch + (long long)r * g.halo_stride
                                : g.band + (long long)i * g.band_pitch + (long long)(r - ny) * g.band_stride;
    else           src = r < 2 * ny ? g.band + (long long)i * g.band_pitch + (long long)(g.band_rows - 2 * ny + r) * g.band_stride
                                    : g.halo[1] + (long long)i * g.halo_pitch + (long long)(r - 2 * ny) * g.halo_stride;
    float *dst = sin + (long long)f * simg + (long long)r * sstride;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < g.cols; c += gridDim.x * 256) dst[c] = src[c];
}

// scatter: the strip's middle ny output rows, columns [c0, c0 + nc), are band rows [0, ny) / [band_rows - ny, band_rows) of the output
__global__ __launch_bounds__(256) void sg2d_strip_scatter_kernel(float *__restrict__ out, int out_stride, long long out_pitch, const float *__restrict__ sout,
                                                                 int sstride, long long simg, const StripGeom g, int c0, int nc, unsigned f0)
{
    const unsigned f = f0 + blockIdx.z, k = f / g.images, i = f - k * g.images;
    const int r = blockIdx.y, side = g.side_of[k];
    const float *src = sout + (long long)f * simg + (long long)(g.ny + r) * sstride;
    float *dst = out + (long long)i * out_pitch + (long long)((side == 0 ? 0 : g.band_rows - g.ny) + r) * out_stride;
    for (int c = c0 + blockIdx.x * 256 + threadIdx.x; c < c0 + nc; c += gridDim.x * 256) dst[c] = src[c];
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

// step (2) alone: the ny output rows next to each artificial edge, from the halo rows and the band's own 2 ny rows next to that edge.
// Both strips of every image are ONE batch of 3 ny-row frames: one gather launch, the filter, one scatter launch.  Gather and filter run on
// `halo_stream` -- the stream the halos arrive on; they read the band and the halos and write scratch only, so they overlap the band launch
// that is still running on `stream` -- and only the scatter of the finished rows is ordered behind `stream`.
static int rowband_edges(const char *who, const Savgol2DFilter *filter, const float *d_band, int band_rows, int cols, int in_stride,
                         size_t in_image_pitch, const float *d_halo_up, const float *d_halo_down, int halo_stride,
                         size_t halo_image_pitch, float *d_out, int out_stride, size_t out_image_pitch, size_t images,
                         Savgol2DBoundary boundary, int method, hipStream_t hst, hipStream_t st)
{
    if (rowband_check(who, filter, d_band, d_out, band_rows, cols, d_halo_up, d_halo_down, halo_stride, halo_image_pitch, images) != 0) return -1;
    if ((!d_halo_up && !d_halo_down) || images == 0) return 0;
    const int nx = filter->config.half_window_x, ny = filter->config.half_window_y;
    if (cols - 2 * nx <= 0 && boundary == SAVGOL2D_BOUNDARY_VALID) { sg_set_error("%s: image smaller than the window", who); return -1; }
    const bool valid = boundary == SAVGOL2D_BOUNDARY_VALID;
    const int srows = 3 * ny, sstride = (cols + 3) & ~3;
    const size_t simg = (size_t)srows * sstride;
    const int nstrips = (d_halo_up ? 1 : 0) + (d_halo_down ? 1 : 0);
    if (images * (size_t)nstrips > 0xffffffffull) { sg_set_error("%s: too many images", who); return -1; }
    sg::DeviceCtx *ctx = sg::ctx_get();
    if (!ctx) return -1;
    float *sin = static_cast<float *>(sg::scratch_alloc(ctx, sizeof(float) * 2 * simg * images * nstrips, hst, "scratch (row-band strips)"));
    if (!sin) return -1;
    float *sout = sin + simg * images * nstrips;
    sg::StripGeom g;
    memset(&g, 0, sizeof(g));
    g.band = d_band; g.band_pitch = (long long)in_image_pitch; g.band_stride = in_stride; g.band_rows = band_rows;
    g.halo[0] = d_halo_up; g.halo[1] = d_halo_down; g.halo_pitch = (long long)halo_image_pitch; g.halo_stride = halo_stride;
    g.ny = ny; g.cols = cols; g.images = (unsigned)images;
    g.side_of[0] = d_halo_up ? 0 : 1; g.side_of[1] = 1;
    const unsigned frames = (unsigned)(images * nstrips);
    unsigned gx = (unsigned)((cols + 255) / 256);
    if (gx > 64) gx = 64;
    int rc = 0;
    for (unsigned f0 = 0; f0 < frames; f0 += 65535u) {
        const unsigned nf = frames - f0 < 65535u ? frames - f0 : 65535u;
        hipLaunchKernelGGL(sg::sg2d_strip_gather_kernel, dim3(gx, (unsigned)srows, nf), dim3(256), 0, hst, sin, sstride, (long long)simg, g, f0);
    }
    if (!sg::hip_ok(hipGetLastError(), "strip gather launch")) rc = -1;
    if (rc == 0 && savgol2d_apply_batch_f32(filter, sin, srows, cols, sstride, simg, sout, sstride, simg, frames, boundary, method, hst) != 0) rc = -1;
    if (rc == 0 && hst != st) {                      // the scatter (and the free) wait for the strips; nothing else on `stream` does
        hipEvent_t ev = nullptr;
        if (!sg::hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate(row-band strips)")) rc = -1;
        else {
            if (!sg::hip_ok(hipEventRecord(ev, hst), "hipEventRecord(row-band strips)") ||
                !sg::hip_ok(hipStreamWaitEvent(st, ev, 0), "hipStreamWaitEvent(row-band strips)")) rc = -1;
            (void)hipEventDestroy(ev);               // released once it has completed
        }
    }
    if (rc == 0) {
        // VALID wrote columns [nx, cols - nx) only
        const int c0 = valid ? nx : 0, nc = valid ? cols - 2 * nx : cols;
        for (unsigned f0 = 0; f0 < frames; f0 += 65535u) {
            const unsigned nf = frames - f0 < 65535u ? frames - f0 : 65535u;
            hipLaunchKernelGGL(sg::sg2d_strip_scatter_kernel, dim3(gx, (unsigned)ny, nf), dim3(256), 0, st, d_out, out_stride, (long long)out_image_pitch, sout,
                               sstride, (long long)simg, g, c0, nc, f0);
        }
        if (!sg::hip_ok(hipGetLastError(), "strip scatter launch")) rc = -1;
    }
    if (rc != 0 && hst != st) (void)hipStreamSynchronize(hst);      // error path: nothing of the strips may still run when the scratch goes back
    if (!sg::scratch_free(sin, st, "scratch free (row-band strips)")) rc = -1;
    return rc;
}

int savgol2d_apply_rowband_edges_f32(const Savgol2DFilter *filter, const float *d_band, int band_rows, int cols, int in_stride,
                                     size_t in_image_pitch, const float *d_halo_up, const float *d_halo_down, int halo_stride,
                                     size_t halo_image_pitch, float *d_out, int out_stride, size_t out_image_pitch, size_t images,
                                     Savgol2DBoundary boundary, int method, void *stream)
{
    return rowband_edges("savgol2d_apply_rowband_edges_f32", filter, d_band, band_rows, cols, in_stride, in_image_pitch, d_halo_up, d_halo_down, halo_stride,
                         halo_image_pitch, d_out, out_stride, out_image_pitch, images, boundary, method, static_cast<hipStream_t>(stream),
                         static_cast<hipStream_t>(stream));
}

int savgol2d_apply_rowband_edges_streams_f32(const Savgol2DFilter *filter, const float *d_band, int band_rows, int cols, int in_stride,
                                             size_t in_image_pitch, const float *d_halo_up, const float *d_halo_down, int halo_stride,
                                             size_t halo_image_pitch, float *d_out, int out_stride, size_t out_image_pitch, size_t images,
                                             Savgol2DBoundary boundary, int method, void *halo_stream, void *stream)
{
    return rowband_edges("savgol2d_apply_rowband_edges_streams_f32", filter, d_band, band_rows, cols, in_stride, in_image_pitch, d_halo_up, d_halo_down,
                         halo_stride, halo_image_pitch, d_out, out_stride, out_image_pitch, images, boundary, method, static_cast<hipStream_t>(halo_stream),
                         static_cast<hipStream_t>(stream));
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
