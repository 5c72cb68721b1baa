// sg_stream_moment_fit.cpp -- host side of the stream block push's block-moment form (csrc/sg_stream_dma.hip, MomTaps): is the bank's tap table a
// polynomial of degree <= 2 in the tap index, and if so, what does a block of 8 consecutive ticks at offset `off` into a window contribute?
// Reference weights: src/savgolFilter.c:151-409 builds the centre row this receives (fp32); the fused bank applies it as one dot product per tick
// (src/savgol_stream.c:25-38).  Same acceptance rule as the 1-D block moments (sg_k1d_moment_fit.cpp): the fitted polynomial must reproduce every
// fp32 tap to 3e-7 of the largest tap, otherwise the bank keeps the tap-by-tap tiles.
#include <cmath>
#include <cstring>

#include "savgol_hip.h"
#include "sg_internal.h"
#include "sg_stream_host.hpp"

namespace sg {

int stream_moment_fit(int n, const float *w, StreamMomentFit *fit)
{
    if (!w || !fit || n < STREAM_MOMENT_MIN_N || n > STREAM_MOMENT_MAX_N) return 0;
    const int K = 2 * n + 1;
    double wmax = 0.0, mean_u2 = 0.0;
    for (int k = 0; k < K; ++k) { wmax = std::fmax(wmax, std::fabs((double)w[k])); mean_u2 += (double)(k - n) * (k - n); }
    mean_u2 /= K;
    if (!(wmax > 0.0)) return 0;
    // orthogonal basis on the symmetric grid u = k - n: 1, u, u^2 - mean(u^2)
    double a[3] = {0, 0, 0}, nrm[3] = {0, 0, 0};
    for (int k = 0; k < K; ++k) {
        const double u = k - n, p[3] = {1.0, u, u * u - mean_u2};
        for (int s = 0; s < 3; ++s) { a[s] += (double)w[k] * p[s]; nrm[s] += p[s] * p[s]; }
    }
    for (int s = 0; s < 3; ++s) a[s] /= nrm[s];
    auto poly = [&](int terms, double u) { double v = a[0]; if (terms > 1) v += a[1] * u; if (terms > 2) v += a[2] * (u * u - mean_u2); return v; };
    int terms = 0;
    for (int m = 1; m <= 3 && !terms; ++m) {
        double worst = 0.0;
        for (int k = 0; k < K; ++k) worst = std::fmax(worst, std::fabs((double)w[k] - poly(m, k - n)));
        if (worst <= 3e-7 * wmax) terms = m;
    }
    if (!terms) return 0;
    memset(fit, 0, sizeof(*fit));
    fit->terms = terms;
    // a block at offset off covers taps off .. off + 7; in the block basis q_0 = 1, q_1 = t - 3.5, q_2 = (t - 3.5)^2 - 5.25 (norms 8, 42, 168)
    for (int off = 0; off <= 2 * n - 7; ++off) {
        double c0 = 0, c1 = 0, c2 = 0;
        for (int t = 0; t < 8; ++t) {
            const double v = poly(terms, off + t - n), q1 = t - 3.5, q2 = q1 * q1 - 5.25;
            c0 += v; c1 += v * q1; c2 += v * q2;
        }
        fit->c[0][off] = (float)(c0 / 8.0);
        fit->c[1][off] = terms > 1 ? (float)(c1 / 42.0) : 0.0f;
        fit->c[2][off] = terms > 2 ? (float)(c2 / 168.0) : 0.0f;
    }
    return terms;
}

}  // namespace sg

extern "C" int savgol_hip_stream_moment_table(int half_window, const float *center_weights, float *coefficients)
{
    if (!center_weights || !coefficients) { sg_set_error("savgol_hip_stream_moment_table: NULL pointer"); return -1; }
    sg::StreamMomentFit fit;
    const int terms = sg::stream_moment_fit(half_window, center_weights, &fit);
    for (int s = 0; s < 3; ++s)
        for (int off = 0; off < SAVGOL_HIP_STREAM_MOMENT_OFFSETS; ++off) coefficients[s * SAVGOL_HIP_STREAM_MOMENT_OFFSETS + off] = terms ? fit.c[s][off] : 0.0f;
    return terms;
}
