// sg_rowband_rccl.hip -- the halo exchanges of the multi-GPU splits over RCCL (xGMI point to point).  OPTIONAL translation unit:
// built into its own library, lib/libsavgol_hip_rccl.so, so that libsavgol_hip.so itself never links librccl.
//
// 2-D row bands (savgol2d_rowband_exchange_rccl): every rank owns a band of rows of every frame (savgol2d_rowband_plan) and needs the
// ny rows next to its band from the neighbour above and below.  1-D length split (savgol_lengthsplit_exchange_rccl): every rank owns a
// segment of every channel and needs the n samples next to each cut.  Both are the same operation on a [outer][rows][cols] block with
// pitches: ONE message per neighbour and direction for the whole stack -- the boundary block is packed into a contiguous buffer by one
// kernel launch per side (round 3 issued 2 x images hipMemcpy2DAsync calls: 1024 launches for config 4's 512 frames), then
//     ncclGroupStart;  ncclSend(a) ncclSend(b) ncclRecv(..) ncclRecv(..);  ncclGroupEnd
// on the caller's stream.  Point to point: one xGMI link per neighbour pair, no all-reduce anywhere (BASELINE config 4 split over 8
// GPUs: 7 rows x 4096 x 4 B x 512 frames = 57 MB per neighbour and direction, ~0.4 ms on a 153 GB/s link -- and the band's own launch
// does not depend on it).  The received rows land in d_halo_up / d_halo_down laid out as savgol2d_apply_rowband_f32 wants them:
// halo_stride = cols, halo_image_pitch = ny * cols.
//
// Executed: tests/test_gpu_rccl_exchange.py creates a one-rank communicator and runs both exchanges against itself (RCCL serves a
// send to the own rank as a local copy; a ring of ONE is the periodic case, so what must arrive where is well defined), and
// bench.py --workload image --rowband times this path (--exchange c, the default when the library loads) on however many GPUs the
// driver gives it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>

#include "savgol_hip_rccl.h"
#include "sg_exchange_order.hpp"

namespace {

// RCCL as the transport of sg::exchange_post (the posting order lives in sg_exchange_order.hpp, where a CPU test can run it too)
struct RcclTransport {
    ncclComm_t comm;
    hipStream_t st;
    bool group_start() { return ncclGroupStart() == ncclSuccess; }
    bool group_end() { return ncclGroupEnd() == ncclSuccess; }
    bool send(const void *p, size_t words, int peer) { return ncclSend(p, words, ncclUint32, peer, comm, st) == ncclSuccess; }
    bool recv(void *p, size_t words, int peer) { return ncclRecv(p, words, ncclUint32, peer, comm, st) == ncclSuccess; }
};

// words = 4-byte units.  Block (o, r) of `cols` words at src + o * pitch + r * stride  ->  dst + (o * rows + r) * cols
__global__ __launch_bounds__(256) void pack_rows_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, long long pitch, long long stride, int rows, int cols)
{
    const long long o = blockIdx.z;
    const int r = blockIdx.y;
    const uint32_t *s = src + o * pitch + (long long)r * stride;
    uint32_t *d = dst + (o * rows + r) * (long long)cols;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < cols; c += gridDim.x * 256) d[c] = s[c];
}

bool pack(uint32_t *dst, const uint32_t *src, long long pitch, long long stride, size_t outer, int rows, int cols, hipStream_t st)
{
    if (rows < 1 || rows > 65535 || cols < 1) return false;          // grid.y is 16-bit (the entry points refuse more rows than that)
    unsigned gx = (unsigned)((cols + 255) / 256);
    if (gx > 64) gx = 64;
    for (size_t o0 = 0; o0 < outer; o0 += 65535) {                   // grid.z is 16-bit too: slices of the outer dimension
        const unsigned no = (unsigned)(outer - o0 < 65535 ? outer - o0 : 65535);
        hipLaunchKernelGGL(pack_rows_kernel, dim3(gx, (unsigned)rows, no), dim3(256), 0, st, dst + (long long)o0 * rows * cols,
                           src + (long long)o0 * pitch, pitch, stride, rows, cols);
    }
    return hipGetLastError() == hipSuccess;
}

// first = the block next to the cut towards peer_a (sent to a), last = the block next to the cut towards peer_b (sent to b).
// recv_a <- what a sends towards us (its LAST block), recv_b <- b's FIRST block; the posting order is sg::exchange_post's (sg_exchange_order.hpp).
int exchange(ncclComm_t comm, int peer_a, int peer_b, const uint32_t *first, const uint32_t *last, long long pitch, long long stride, size_t outer,
             int rows, int cols, uint32_t *recv_a, uint32_t *recv_b, uint32_t *scratch, hipStream_t st)
{
    const size_t per_side = outer * (size_t)rows * (size_t)cols;
    const bool a = peer_a >= 0, b = peer_b >= 0;
    if ((a && !recv_a) || (b && !recv_b) || ((a || b) && !scratch)) return -1;
    uint32_t *send_a = scratch, *send_b = scratch + per_side;
    if (a && !pack(send_a, first, pitch, stride, outer, rows, cols, st)) return -1;
    if (b && !pack(send_b, last, pitch, stride, outer, rows, cols, st)) return -1;
    RcclTransport t{comm, st};
    return sg::exchange_post(t, peer_a, peer_b, send_a, send_b, recv_a, recv_b, per_side);
}

}  // namespace

extern "C" {

int savgol2d_rowband_exchange_rccl_peers(void *nccl_comm, int peer_up, int peer_down, const float *d_band, int band_rows, int cols, int in_stride,
                                         size_t in_image_pitch, size_t images, int half_win_y, float *d_halo_up, float *d_halo_down,
                                         float *d_send_scratch, void *stream)
{
    if (!nccl_comm || !d_band || half_win_y < 1 || band_rows < half_win_y || cols <= 0 || in_stride < cols || images == 0) return -1;
    const uint32_t *band = reinterpret_cast<const uint32_t *>(d_band);
    return exchange(static_cast<ncclComm_t>(nccl_comm), peer_up, peer_down, band, band + (size_t)(band_rows - half_win_y) * (size_t)in_stride,
                    (long long)in_image_pitch, in_stride, images, half_win_y, cols, reinterpret_cast<uint32_t *>(d_halo_up),
                    reinterpret_cast<uint32_t *>(d_halo_down), reinterpret_cast<uint32_t *>(d_send_scratch), static_cast<hipStream_t>(stream));
}

int savgol2d_rowband_exchange_rccl(void *nccl_comm, int rank, int world_size, const float *d_band, int band_rows, int cols, int in_stride,
                                   size_t in_image_pitch, size_t images, int half_win_y, float *d_halo_up, float *d_halo_down,
                                   float *d_send_scratch, void *stream)
{
    if (rank < 0 || rank >= world_size) return -1;
    return savgol2d_rowband_exchange_rccl_peers(nccl_comm, rank > 0 ? rank - 1 : -1, rank + 1 < world_size ? rank + 1 : -1, d_band, band_rows, cols,
                                                in_stride, in_image_pitch, images, half_win_y, d_halo_up, d_halo_down, d_send_scratch, stream);
}

int savgol_lengthsplit_exchange_rccl(void *nccl_comm, int peer_prev, int peer_next, const void *d_segment, size_t channels, size_t own, size_t ld,
                                     int half_window, int elem_bytes, void *d_halo_prev, void *d_halo_next, void *d_send_scratch, void *stream)
{
    // (more than 65 535 channels: shard by channel, not by length -- one launch dimension carries the channels)
    if (!nccl_comm || !d_segment || half_window < 1 || own < (size_t)half_window || ld < own || channels == 0 || channels > 65535 ||
        (elem_bytes != 4 && elem_bytes != 8)) return -1;
    const int wpe = elem_bytes / 4;                                  // 4-byte words per sample
    const uint32_t *seg = static_cast<const uint32_t *>(d_segment);
    // one "image" of `channels` rows, each row the n samples next to a cut
    return exchange(static_cast<ncclComm_t>(nccl_comm), peer_prev, peer_next, seg, seg + (own - (size_t)half_window) * wpe, 0, (long long)ld * wpe, 1,
                    (int)channels, half_window * wpe, static_cast<uint32_t *>(d_halo_prev), static_cast<uint32_t *>(d_halo_next),
                    static_cast<uint32_t *>(d_send_scratch), static_cast<hipStream_t>(stream));
}

}  // extern "C"
