// sg_k1d_inst.hip -- instantiates sg1d_center_kernel<SG_T, N, V> (narrow tile, and the wide one where it differs) for N in [SG_NLO, SG_NHI] and exports
// one launcher for that group.  Compiled several times by the Makefile (one object per group) so
// the 64 heavily unrolled instantiations build in parallel.
#include "sg_k1d.hpp"

#include <cstdio>
#include <cstdlib>

#if !defined(SG_T) || !defined(SG_NLO) || !defined(SG_NHI) || !defined(SG_FN)
#error "compile with -DSG_T=float|double -DSG_NLO=.. -DSG_NHI=.. -DSG_FN=symbol"
#endif

namespace sg {

template <typename T, int N, int HI>
struct Dispatch1D {
    template <int V>
    static void launch(const Job1D &job, const Taps &taps, unsigned grid, hipStream_t st)
    {
        static const bool debug = getenv("SAVGOL_HIP_DEBUG") != nullptr;
        if (debug) {
            int nb = -1;
            hipFuncAttributes fa;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sg1d_center_kernel<T, N, V>, 256, 0);
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(sg1d_center_kernel<T, N, V>));
            fprintf(stderr, "[savgol-hip] sg1d_center_kernel<%s,%d,%d>: grid %u x 256, occupancy API %d blocks/CU, %d VGPR, %zu B LDS\n",
                    sizeof(T) == 4 ? "float" : "double", N, V, grid, nb, fa.numRegs, fa.sharedSizeBytes);
        }
        hipLaunchKernelGGL((sg1d_center_kernel<T, N, V>), dim3(grid * (4 / SG_K1D_WAVES)), dim3(64 * SG_K1D_WAVES), 0, st, job, taps);     // `grid` counts blocks of 4 tiles
    }
    // wide != 0: the job's tiles were laid out with wide_vectors_per_lane (the host only asks where that differs from the narrow tile)
    static int go(int n, int wide, const Job1D &job, const Taps &taps, unsigned grid, hipStream_t st)
    {
        if (n == N) {
            constexpr int VN = vectors_per_lane(sizeof(T), N), VW = wide_vectors_per_lane(sizeof(T), N);
            if constexpr (VW != VN) { if (wide) { launch<VW>(job, taps, grid, st); return 1; } }
            launch<VN>(job, taps, grid, st);
            return 1;
        }
        if constexpr (N < HI) return Dispatch1D<T, N + 1, HI>::go(n, wide, job, taps, grid, st);
        else return 0;
    }
};

#ifdef SG_FN_STRIDED
template <int N, int HI>
static int dispatch_strided(int n, const JobStrided &job, const Taps &taps, unsigned grid, hipStream_t st)
{
    if (n == N) { hipLaunchKernelGGL((sg1d_strided_kernel<N>), dim3(grid * (4 / SG_K1D_WAVES)), dim3(64 * SG_K1D_WAVES), 0, st, job, taps); return 1; }
    if constexpr (N < HI) return dispatch_strided<N + 1, HI>(n, job, taps, grid, st);
    else return 0;
}
#endif

}  // namespace sg

#ifdef SG_FN_STRIDED
// the fused strided kernel of this group's half windows (fp32 objects only); 1 if this group owns n
extern "C" int SG_FN_STRIDED(int n, const sg::JobStrided *job, const sg::Taps *taps, unsigned grid, void *stream)
{
    return sg::dispatch_strided<SG_NLO, SG_NHI>(n, *job, *taps, grid, static_cast<hipStream_t>(stream));
}
#endif

// returns 1 if this group owns half window n (kernel enqueued), 0 otherwise
extern "C" int SG_FN(int n, int wide, const sg::Job1D *job, const sg::Taps *taps, unsigned grid, void *stream)
{
    return sg::Dispatch1D<SG_T, SG_NLO, SG_NHI>::go(n, wide, *job, *taps, grid, static_cast<hipStream_t>(stream));
}
