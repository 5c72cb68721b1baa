// sg_rowband_rccl.cpp -- the halo exchange of the 2-D row-band split over RCCL (xGMI point to point).  OPTIONAL translation unit:
// built into its own library, lib/libsavgol_hip_rccl.so, so that libsavgol_hip.so itself never links librccl.
//
// Every rank owns a band of rows of every frame (savgol2d_rowband_plan) and needs the ny rows next to its band from the
// neighbour above and below.  One message per neighbour and direction for the WHOLE stack of frames: the boundary rows are
// packed into a contiguous [images][ny][cols] buffer (one 2-D copy per side), then
//     ncclGroupStart;  ncclSend(up) ncclRecv(up) ncclSend(down) ncclRecv(down);  ncclGroupEnd
// on the caller's stream.  Point to point: one xGMI link per neighbour pair, no all-reduce anywhere (BASELINE config 4 split
// over 8 GPUs: 7 rows x 4096 x 4 B x 512 frames = 57 MB per neighbour and direction, ~0.4 ms on a 153 GB/s link -- and
// savgol2d_apply_rowband_f32's first launch, the band itself, does not depend on it).  The received rows land in
// d_halo_up / d_halo_down, laid out as savgol2d_apply_rowband_f32 wants them: halo_stride = cols, halo_image_pitch = ny * cols.
//
// Never executed on this project's single-GPU test boxes (RCCL refuses two ranks on one device); the arithmetic side is covered
// by tests that copy the halos device to device.  Compiled here so that it is known to build against the installed RCCL.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstddef>
#include <cstdio>

#include "savgol_hip_rccl.h"

extern "C" int savgol2d_rowband_exchange_rccl(void *nccl_comm, int rank, int world_size, const float *d_band, int band_rows, int cols,
                                              int in_stride, size_t in_image_pitch, size_t images, int half_win_y, float *d_halo_up,
                                              float *d_halo_down, float *d_send_scratch, void *stream)
{
    if (!nccl_comm || !d_band || !d_send_scratch || rank < 0 || rank >= world_size || half_win_y < 1 || band_rows < half_win_y || cols <= 0) return -1;
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ny = half_win_y;
    const size_t per_side = images * (size_t)ny * (size_t)cols;
    const bool up = rank > 0, down = rank + 1 < world_size;
    if ((up && !d_halo_up) || (down && !d_halo_down)) return -1;
    float *send_up = d_send_scratch, *send_down = d_send_scratch + per_side;
    // pack: rows [0, ny) / [band_rows - ny, band_rows) of every image -> contiguous.  One image's ny rows are ny * cols floats with
    // row pitch in_stride: a 2-D copy per image would be `images` calls, so treat (image, row) as the outer dimension when the
    // frames are dense (in_image_pitch a multiple of in_stride is not required: fall back to per-image copies otherwise).
    for (size_t k = 0; k < images; ++k) {
        if (up && hipMemcpy2DAsync(send_up + k * (size_t)ny * cols, sizeof(float) * cols, d_band + k * in_image_pitch, sizeof(float) * in_stride,
                                   sizeof(float) * cols, ny, hipMemcpyDeviceToDevice, st) != hipSuccess) return -1;
        if (down && hipMemcpy2DAsync(send_down + k * (size_t)ny * cols, sizeof(float) * cols,
                                     d_band + k * in_image_pitch + (size_t)(band_rows - ny) * in_stride, sizeof(float) * in_stride,
                                     sizeof(float) * cols, ny, hipMemcpyDeviceToDevice, st) != hipSuccess) return -1;
    }
    if (ncclGroupStart() != ncclSuccess) return -1;
    bool ok = true;
    if (up) ok = ok && ncclSend(send_up, per_side, ncclFloat, rank - 1, comm, st) == ncclSuccess &&
                 ncclRecv(d_halo_up, per_side, ncclFloat, rank - 1, comm, st) == ncclSuccess;
    if (down) ok = ok && ncclSend(send_down, per_side, ncclFloat, rank + 1, comm, st) == ncclSuccess &&
                   ncclRecv(d_halo_down, per_side, ncclFloat, rank + 1, comm, st) == ncclSuccess;
    if (ncclGroupEnd() != ncclSuccess) return -1;
    return ok ? 0 : -1;
}
