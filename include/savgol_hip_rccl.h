/*
 * savgol_hip_rccl.h -- the OPTIONAL RCCL side of the 2-D row-band split: lib/libsavgol_hip_rccl.so (csrc/sg_rowband_rccl.cpp),
 * the only part of the project that links librccl.  libsavgol_hip.so itself never does.
 *
 * Reference: none -- the reference (Tugbars/Savitzky-Golay-Filter) has no multi-device path; this fills the halo buffers
 * savgol2d_apply_rowband_f32 (savgol_hip.h) takes, so that a band's rows equal the rows the reference's whole-frame call
 * (savgol2d_apply, src/savgol2d.c:398-456) produces.
 */
#ifndef SAVGOL_HIP_RCCL_H
#define SAVGOL_HIP_RCCL_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One nearest-neighbour exchange for a whole stack of frames, enqueued on `stream`:
 *   nccl_comm        an ncclComm_t of `world_size` ranks, this process being `rank` (band order = rank order)
 *   d_band           this rank's band: band_rows x cols per image, in_stride / in_image_pitch in elements
 *   d_halo_up/_down  receive buffers, images x half_win_y x cols floats each (NULL allowed where there is no neighbour);
 *                    laid out for savgol2d_apply_rowband_f32 with halo_stride = cols, halo_image_pitch = half_win_y * cols
 *   d_send_scratch   2 x images x half_win_y x cols floats (the packed boundary rows; must stay untouched until the stream passes)
 * ncclGroupStart; ncclSend / ncclRecv to rank - 1 and rank + 1; ncclGroupEnd -- point to point, one xGMI link per neighbour pair.
 * 0 on success, -1 on bad arguments or an RCCL / HIP error.                                                              */
int savgol2d_rowband_exchange_rccl(void *nccl_comm, int rank, int world_size, const float *d_band, int band_rows, int cols,
                                   int in_stride, size_t in_image_pitch, size_t images, int half_win_y, float *d_halo_up,
                                   float *d_halo_down, float *d_send_scratch, void *stream);

/* The same with the two peers named by the caller (-1 = no neighbour on that side) instead of rank - 1 / rank + 1.  When both peers are
 * the SAME rank -- a ring of two, or a single rank exchanging with itself, which is how tests/test_gpu_rccl_exchange.py executes this on
 * one GPU -- the frame is a ring: d_halo_up receives the peer's LAST half_win_y rows and d_halo_down its FIRST ones.                    */
int savgol2d_rowband_exchange_rccl_peers(void *nccl_comm, int peer_up, int peer_down, const float *d_band, int band_rows, int cols,
                                         int in_stride, size_t in_image_pitch, size_t images, int half_win_y, float *d_halo_up,
                                         float *d_halo_down, float *d_send_scratch, void *stream);

/* The 1-D length split's exchange (savitzky-golay-filter_amd/lengthsplit.py; reference loop served: src/savgolFilter.c:763-766 on a
 * channel cut across ranks): every rank owns samples [lo, hi) of every channel -- d_segment = channels rows of `own` samples, ld apart,
 * elem_bytes 4 (fp32) or 8 (fp64) -- sends its first half_window samples of every channel to peer_prev and its last ones to peer_next
 * and receives the neighbours' into d_halo_prev / d_halo_next (channels x half_window samples each, contiguous).  -1 = no neighbour;
 * PERIODIC signals close the ring (rank 0's prev = the last rank), and with two ranks or one both peers are the same rank: d_halo_prev
 * then receives that peer's LAST samples and d_halo_next its FIRST.  d_send_scratch: 2 x channels x half_window samples.  At most
 * 65 535 channels per call (more: shard by channel, which needs no exchange).  0 / -1.                                                 */
int savgol_lengthsplit_exchange_rccl(void *nccl_comm, int peer_prev, int peer_next, const void *d_segment, size_t channels, size_t own,
                                     size_t ld, int half_window, int elem_bytes, void *d_halo_prev, void *d_halo_next,
                                     void *d_send_scratch, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SAVGOL_HIP_RCCL_H */
