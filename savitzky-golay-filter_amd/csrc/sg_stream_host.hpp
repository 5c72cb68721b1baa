// sg_stream_host.hpp -- host-only declarations of the stream block push (no device types: included by g++ translation units too)
#pragma once

namespace sg {

// Block-moment form of the fused bank's LDS-DMA tiles (sg_stream_dma.hip, MomTaps): half windows it is built for, and the host fit's result
constexpr int STREAM_MOMENT_MIN_N = 12;                      // below, a window holds at most two whole 8-tick blocks: nothing to win
constexpr int STREAM_MOMENT_MAX_N = 20;                      // interleaved A/B on config 3's shape (profiles/r05_stream_moment_ab.txt): n = 12 / 14 / 16 10 % ahead of the
                                                             // tap-by-tap tiles, 20 level, 24 / 28 / 32 behind by 5 / 9 / 2 % (their 2n halo rows want the waves the moment tiles give up)
constexpr int STREAM_MOMENT_OFFSETS = 2 * STREAM_MOMENT_MAX_N - 6;   // block offsets 0 .. 2n - 7
struct StreamMomentFit {
    int   terms;                                             // 1..3 moments per block (polynomial degree + 1)
    float c[3][STREAM_MOMENT_OFFSETS];                       // c[s][off]: weight of moment s of the block that starts `off` taps into the window
};
// 0 = the table is not a polynomial of degree <= 2 to 3e-7 of its largest tap (or n outside the range): keep the tap-by-tap tiles
int stream_moment_fit(int n, const float *center_weights, StreamMomentFit *fit);

}  // namespace sg
