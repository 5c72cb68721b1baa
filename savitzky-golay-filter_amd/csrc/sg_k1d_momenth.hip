// sg_k1d_momenth.hip -- instantiates the fp32 half-lane block-moment kernel (sg_k1d_momenth.hpp) for half windows 24..32 and ONE moment count per
// object (SG_MOMENT_TERMS = 3, 5 or 7; built three times by the Makefile), and exports its launcher.
// Two waves per block for this kernel family (the others: four): over six placements of the headline's two buffers the same tiles in blocks of two are
// 0.6-1.4 % faster on every one (5.427 against 5.478 ms median; one wave per block: 5.40 median but 5.30 ... 5.72 -- tools/placement_1d.py, profiles/
// r05_placement_1d.txt).  The XCD chunk stays 2 MiB of input: sg1d_tile_body scales its block count by 4 / WAVES.
#define SG_K1D_WAVES 2
#include "sg_k1d_momenth.hpp"

#include <cstdio>
#include <cstdlib>

#if !defined(SG_MOMENT_TERMS) || !defined(SG_MOMENT_FN)
#error "compile with -DSG_MOMENT_TERMS=3|5|7 -DSG_MOMENT_FN=symbol"
#endif

namespace sg {

template <int N>
static int launch_momenth(int n, const Job1D &job, const MomentArgs &args, unsigned grid, hipStream_t st)
{
    if (n == N) {
        hipLaunchKernelGGL((sg1d_center_momenth_kernel<N, SG_MOMENT_TERMS>), dim3(grid * (4 / SG_K1D_WAVES)), dim3(64 * SG_K1D_WAVES), 0, st, job, args);     // `grid` counts blocks of 4 tiles
        return 0;
    }
    if constexpr (N < MOMENT_MAX_N) return launch_momenth<N + 1>(n, job, args, grid, st);
    else return 1;
}

}  // namespace sg

extern "C" int SG_MOMENT_FN(int n, const sg::Job1D *job, const float *d_table, unsigned grid, void *stream)
{
    const sg::MomentArgs args{d_table};
    static const bool debug = getenv("SAVGOL_HIP_DEBUG") != nullptr;
    if (debug) fprintf(stderr, "[savgol-hip] sg1d_center_momenth_kernel<%d,%d>: grid %u x 256\n", n, SG_MOMENT_TERMS, grid);
    if (sg::launch_momenth<sg::MOMENTH_MIN_N>(n, *job, args, grid, static_cast<hipStream_t>(stream)) != 0) {
        sg_set_error("no half-lane moment kernel for half_window %d", n);
        return -1;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sg_set_error("1-D half-lane moment kernel launch failed: %s", hipGetErrorString(e)); return -1; }
    return 0;
}
