// sg_k1d_moment_fit.cpp -- host side of the block-moment kernels (sg_k1d_momenth.hpp: fp32, half windows 20..32; sg_k1d_moment64.hpp: fp64,
// 24..32): recover the polynomial behind a filter's 2n+1 fp32 centre taps and re-express the taps that fall on a group's common block in that
// block's Legendre basis.  Double
// precision throughout; the results are rounded to fp32 once, when the table is written.
//
// The taps are w[k] = p(k) + rounding, p of degree <= poly_order (reference compute_weight, src/savgolFilter.c:336-356).
// Nothing here trusts that: the degree is found by fitting (3, 5, then 7 terms) and the fit must reproduce every tap to
// 3e-7 of the largest one -- the size of the fp32 rounding already in the reference's table -- or the caller keeps the
// plain kernel.  Hand-edited tables therefore still run, just not on this path.
#include <cmath>
#include <cstring>

#include "sg_k1d_host.hpp"

namespace {

constexpr int MAXWS = 65, MAXBLOCK = 32, MAXT = sg::MOMENT_MAX_TERMS;

void legendre(double z, int terms, double *P)          // P[s] = P_s(z), s < terms
{
    P[0] = 1.0;
    if (terms > 1) P[1] = z;
    for (int s = 2; s < terms; ++s) P[s] = ((2 * s - 1) * z * P[s - 1] - (s - 1) * P[s - 2]) / s;
}

// solve G x = b (G symmetric positive definite, terms x terms) by Gaussian elimination with partial pivoting
bool solve(int terms, const double (*G)[MAXT], const double *b, double *x)
{
    double a[MAXT][MAXT + 1];
    for (int i = 0; i < terms; ++i) { for (int j = 0; j < terms; ++j) a[i][j] = G[i][j]; a[i][terms] = b[i]; }
    for (int c = 0; c < terms; ++c) {
        int piv = c;
        for (int r = c + 1; r < terms; ++r) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (std::fabs(a[piv][c]) < 1e-300) return false;
        if (piv != c) for (int j = 0; j <= terms; ++j) { const double t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t; }
        for (int r = c + 1; r < terms; ++r) {
            const double f = a[r][c] / a[c][c];
            for (int j = c; j <= terms; ++j) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = terms - 1; i >= 0; --i) {
        double s = a[i][terms];
        for (int j = i + 1; j < terms; ++j) s -= a[i][j] * x[j];
        x[i] = s / a[i][i];
    }
    return true;
}

// least-squares coefficients of y[0..count) on P_s(z_i), s < terms
bool fit(int terms, int count, const double *z, const double *y, double *coef)
{
    double G[MAXT][MAXT] = {}, b[MAXT] = {}, P[MAXT];
    for (int i = 0; i < count; ++i) {
        legendre(z[i], terms, P);
        for (int s = 0; s < terms; ++s) { b[s] += P[s] * y[i]; for (int t = 0; t < terms; ++t) G[s][t] += P[s] * P[t]; }
    }
    return solve(terms, G, b, coef);
}

// The polynomial behind a filter's centre taps: degree found by fitting 3, 5, then 7 Legendre terms; terms == 0: the table is not such a polynomial
// (to 3e-7 of its largest tap) and the caller keeps the plain kernel.  One fit for both block-moment kernels (it used to be
// pasted into each prepare function).
struct CentreFit {
    int terms = 0, n = 0;
    double coef[MAXT] = {};
    double at(double k) const                     // the polynomial at a real tap index
    {
        double P[MAXT], v = 0.0;
        legendre((k - n) / n, terms, P);
        for (int s = 0; s < terms; ++s) v += coef[s] * P[s];
        return v;
    }
};
CentreFit fit_centre_taps(int n, const float *w)
{
    CentreFit f;
    f.n = n;
    const int WS = 2 * n + 1;
    double zk[MAXWS], wk[MAXWS], wmax = 0.0;
    for (int k = 0; k < WS; ++k) {
        zk[k] = (double)(k - n) / n; wk[k] = (double)w[k];
        if (!std::isfinite(wk[k])) return f;
        if (std::fabs(wk[k]) > wmax) wmax = std::fabs(wk[k]);
    }
    if (wmax == 0.0) return f;
    for (int t : {3, 5, 7}) {
        double c[MAXT] = {}, P[MAXT];
        if (!fit(t, WS, zk, wk, c)) continue;
        double worst = 0.0;
        for (int k = 0; k < WS; ++k) {
            legendre(zk[k], t, P);
            double v = 0.0;
            for (int s = 0; s < t; ++s) v += c[s] * P[s];
            worst = std::fmax(worst, std::fabs(v - wk[k]));
        }
        if (worst <= 3e-7 * wmax) { f.terms = t; memcpy(f.coef, c, sizeof(f.coef)); break; }
    }
    return f;
}

}  // namespace

// The fp32 half-lane form (sg_k1d_momenth.hpp): 16 outputs per group, the block X[LO .. HI) of the group's window, front / back pairing.
extern "C" int sg1d_momenth_prepare(int n, const float *w, float *table)
{
    if (n < sg::MOMENTH_MIN_N || n > sg::MOMENT_MAX_N) return 0;
    const int WS = 2 * n + 1, OFF = sg::moment_off(n), LO = sg::momenth_lo(n), HI = sg::momenth_hi(n), BLOCK = HI - LO;
    const CentreFit cf = fit_centre_taps(n, w);
    if (!cf.terms) return 0;
    const int terms = cf.terms;
    auto p = [&](double k) { return cf.at(k); };
    memset(table, 0, sizeof(float) * sg::MOMENT_TABLE_FLOATS);
    for (int k = 0; k <= WS; ++k) {                               // pair k = (w[k], w[k-1]); w[-1] = w[2n+1] = 0
        table[sg::MOMENTH_OFF_W + 2 * k] = k < WS ? w[k] : 0.0f;
        table[sg::MOMENTH_OFF_W + 2 * k + 1] = k >= 1 ? w[k - 1] : 0.0f;
    }
    // block basis, rounded to fp32 as the kernel will use it; the kernel forms phi_s(BLOCK-1-t) as (-1)^s phi_s(t)
    float phi[2 * sg::MOMENTH_MAX_STEPS * 2][MAXT];
    for (int t = 0; t < BLOCK / 2; ++t) {
        double P[MAXT];
        legendre((t - 0.5 * (BLOCK - 1)) / (0.5 * BLOCK), terms, P);
        for (int s = 0; s < terms; ++s) phi[t][s] = (float)P[s];
    }
    for (int u = 0; u < BLOCK / 4; ++u)
        for (int s = 1; s < terms; ++s) {
            table[sg::MOMENTH_OFF_PHI + (u * 6 + (s - 1)) * 2] = phi[2 * u][s];
            table[sg::MOMENTH_OFF_PHI + (u * 6 + (s - 1)) * 2 + 1] = phi[2 * u + 1][s];
        }
    for (int r = 0; r < 16; ++r) {
        double G[MAXT][MAXT] = {}, b[MAXT] = {}, c[MAXT] = {};
        for (int t = 0; t < BLOCK; ++t) {
            double Pr[MAXT];
            const int tm = t < BLOCK / 2 ? t : BLOCK - 1 - t;
            Pr[0] = 1.0;
            for (int s = 1; s < terms; ++s) Pr[s] = (t < BLOCK / 2 || !(s & 1)) ? (double)phi[tm][s] : -(double)phi[tm][s];
            const double q = p((double)(LO + t - r - OFF));
            for (int s = 0; s < terms; ++s) { b[s] += Pr[s] * q; for (int u = 0; u < terms; ++u) G[s][u] += Pr[s] * Pr[u]; }
        }
        if (!solve(terms, G, b, c)) return 0;
        for (int s = 0; s < terms; ++s) table[sg::MOMENTH_OFF_C + s * 16 + r] = (float)c[s];
    }
    return terms;
}

// The fp64 kernel (sg_k1d_moment64.hpp; savgol_apply_batch_f64_tol / SAVGOL_BATCH_MOMENT_F64): 16 outputs per lane, the block X[LO .. HI) = 2n - 14 samples.  Everything
// stays in double: the taps applied one by one are exact promotions of the fp32 table, the block's share comes from the fitted polynomial.
extern "C" int sg1d_moment64_prepare(int n, const float *w, double *table)
{
    if (n < sg::MOMENT_MIN_N || n > sg::MOMENT_MAX_N) return 0;
    const int OFF = sg::moment64_off(n), LO = sg::moment64_lo(n), HI = sg::moment64_hi(n), BLOCK = HI - LO;
    const CentreFit cf = fit_centre_taps(n, w);
    if (!cf.terms) return 0;
    const int terms = cf.terms;
    auto p = [&](double k) { return cf.at(k); };
    memset(table, 0, sizeof(double) * sg::MOMENT64_TABLE_DOUBLES);
    for (int k = 0; k < 15; ++k) table[sg::MOMENT64_OFF_W + k] = (double)w[k];
    double phi[2 * sg::MOMENT64_MAX_PAIRS][MAXT];
    for (int t = 0; t < BLOCK; ++t) legendre((t - 0.5 * (BLOCK - 1)) / (0.5 * BLOCK), terms, phi[t]);
    for (int t = 0; t < BLOCK / 2; ++t)
        for (int s = 1; s < terms; ++s) table[sg::MOMENT64_OFF_PHI + t * 6 + (s - 1)] = phi[t][s];
    for (int r = 0; r < 16; ++r) {
        double G[MAXT][MAXT] = {}, b[MAXT] = {}, c[MAXT] = {};
        for (int t = 0; t < BLOCK; ++t) {
            // the kernel forms phi_s(BLOCK-1-t) as (-1)^s phi_s(t): use exactly those values
            double Pr[MAXT];
            const int tm = t < BLOCK / 2 ? t : BLOCK - 1 - t;
            for (int s = 0; s < terms; ++s) Pr[s] = (t < BLOCK / 2 || !(s & 1)) ? phi[tm][s] : -phi[tm][s];
            const double q = p((double)(LO + t - r - OFF));
            for (int s = 0; s < terms; ++s) { b[s] += Pr[s] * q; for (int u = 0; u < terms; ++u) G[s][u] += Pr[s] * Pr[u]; }
        }
        if (!solve(terms, G, b, c)) return 0;
        for (int s = 0; s < terms; ++s) table[sg::MOMENT64_OFF_C + s * 16 + r] = c[s];
    }
    return terms;
}

