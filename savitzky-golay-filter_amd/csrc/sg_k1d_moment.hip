// sg_k1d_moment.hip -- instantiates the half_window = 32 fp32 kernel that replaces the 32 taps falling on a lane's own block by
// block moments (sg_k1d_moment.hpp) for 3, 5 and 7 moments (poly_order <= 2, <= 4, <= 6), and exports its launcher.
#include "sg_k1d_moment.hpp"

#include <cstdio>
#include <cstdlib>

extern "C" int sg1d_launch_f32_moment(int terms, const sg::Job1D *job, const float *d_table, unsigned grid, void *stream)
{
    const hipStream_t st = static_cast<hipStream_t>(stream);
    const sg::MomentArgs args{d_table};
    static const bool debug = getenv("SAVGOL_HIP_DEBUG") != nullptr;
    if (debug) fprintf(stderr, "[savgol-hip] sg1d_center_moment_kernel<%d>: grid %u x 256\n", terms, grid);
    switch (terms) {
    case 3: hipLaunchKernelGGL((sg::sg1d_center_moment_kernel<3>), dim3(grid), dim3(256), 0, st, *job, args); break;
    case 5: hipLaunchKernelGGL((sg::sg1d_center_moment_kernel<5>), dim3(grid), dim3(256), 0, st, *job, args); break;
    case 7: hipLaunchKernelGGL((sg::sg1d_center_moment_kernel<7>), dim3(grid), dim3(256), 0, st, *job, args); break;
    default: sg_set_error("no moment kernel with %d terms", terms); return -1;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sg_set_error("1-D moment kernel launch failed: %s", hipGetErrorString(e)); return -1; }
    return 0;
}
