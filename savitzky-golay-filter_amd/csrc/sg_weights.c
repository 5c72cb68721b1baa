/*
 * sg_weights.c -- host-side weight tables (1-D Gram-polynomial rows, 2-D least-squares kernel).
 *
 * These tables are tiny (<= 33 x 65 floats, <= 33 x 33 floats) and are built once per filter on
 * the host, "exactly as the reference does": the arithmetic below reproduces the reference's
 * operation order in fp32 (1-D) and fp64 (2-D) so the tables are bit-identical to
 *   src/savgolFilter.c:151-176 (falling factorials), :236-303 (Gram recurrence),
 *   :336-409 (weights), :707 (dt_scale)                    and
 *   src/savgol2d.c:77-265 (design matrix, normal equations, Cholesky, pinv row), :321-322 (scale).
 * The structure is different: instead of re-running the recurrence for every (tap, target, order)
 * triple, the Gram values are tabulated once per abscissa and shared by all rows.
 *
 * MUST be compiled with -ffp-contract=off (see the Makefile): a fused multiply-add anywhere in
 * here changes the last bit of some weights.
 */
#include "sg_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ *
 * 1-D                                                                                        *
 * ------------------------------------------------------------------------------------------ */

/* a (a-1) ... (a-b+1), product carried in double and rounded to float once */
static float falling(int a, int b)
{
    double p = 1.0;
    if (b > a) return 0.0f;
    for (int j = a - b + 1; j <= a; ++j) p *= (double)j;   /* ascending, as the reference: long products round */
    return (float)p;
}

/* Gram values of all orders at one abscissa x, for one derivative order s_max:
 * g[k*stride + s] = F_k^{(s)}(x).  Orders run 0..m. */
static void gram_column(int n, int m, int s_max, int x, float *g, int stride)
{
    const float nf = (float)n, xf = (float)x;
    for (int s = 0; s <= s_max; ++s) g[s] = (s == 0) ? 1.0f : 0.0f;
    if (m >= 1) {
        const float rn = 1.0f / nf;
        float *g1 = g + stride;
        g1[0] = rn * (xf * g[0]);
        for (int s = 1; s <= s_max; ++s) g1[s] = rn * (xf * g[s] + (float)s * g[s - 1]);
    }
    const float n2 = 2.0f * nf;
    for (int k = 2; k <= m; ++k) {
        const float kf = (float)k;
        const float den = kf * (n2 - kf + 1.0f);
        const float alpha = (4.0f * kf - 2.0f) / den;
        const float gamma = ((kf - 1.0f) * (n2 + kf)) / den;
        const float *p1 = g + (size_t)(k - 1) * stride, *p2 = g + (size_t)(k - 2) * stride;
        float *c = g + (size_t)k * stride;
        c[0] = alpha * (xf * p1[0]) - gamma * p2[0];
        for (int s = 1; s <= s_max; ++s) {
            const float inner = xf * p1[s] + (float)s * p1[s - 1];
            c[s] = alpha * inner - gamma * p2[s];
        }
    }
}

int sg_weights_valid(int n, int m, int d, float time_step)
{
    if (n < 1 || n > SAVGOL_MAX_HALF_WINDOW) return 0;
    if (m < 0 || m >= 2 * n + 1) return 0;
    if (d < 0 || d > SAVGOL_MAX_DERIVATIVE || d > m) return 0;
    if (!(time_step > 0.0f)) return 0;
    if (2 * n + m + 1 >= 76) return 0;          /* the reference's 76-entry factorial table */
    return 1;
}

void sg_weights_fill(SavgolFilter *f)
{
    const int n = f->config.half_window, m = f->config.poly_order, d = f->config.derivative;
    const int ws = 2 * n + 1;
    enum { S = SAVGOL_MAX_DERIVATIVE + 1, K = SAVGOL_MAX_WINDOW };

    /* value table: tap abscissae -n..n, derivative order 0 only            -> tap_g[i][k]
     * target table: evaluation points t = 0..n, derivative order d only    -> tgt_g[t][k]   */
    static _Thread_local float tap_g[SAVGOL_MAX_WINDOW][K];
    static _Thread_local float tgt_g[SAVGOL_MAX_HALF_WINDOW + 1][K];
    float col[K * S];
    float norm[K];

    for (int k = 0; k <= m; ++k)
        norm[k] = (float)(2 * k + 1) * (falling(2 * n, k) / falling(2 * n + k + 1, k + 1));

    for (int i = 0; i < ws; ++i) {
        gram_column(n, m, 0, i - n, col, 1);
        for (int k = 0; k <= m; ++k) tap_g[i][k] = col[k];
    }
    for (int t = 0; t <= n; ++t) {
        gram_column(n, m, d, t, col, S);
        for (int k = 0; k <= m; ++k) tgt_g[t][k] = col[k * S + d];
    }

    f->window_size = ws;
    f->dt_scale = powf(f->config.time_step, (float)d);

    for (int row = -1; row < n; ++row) {               /* row -1 = centre (t = 0), row e -> t = n-e */
        const int t = (row < 0) ? 0 : n - row;
        float *dst = (row < 0) ? f->center_weights : f->edge_weights[row];
        for (int i = 0; i < ws; ++i) {
            float w = 0.0f;
            for (int k = 0; k <= m; ++k) w += norm[k] * tap_g[i][k] * tgt_g[t][k];
            dst[i] = w;
        }
    }
}

/* ------------------------------------------------------------------------------------------ *
 * 2-D                                                                                        *
 * ------------------------------------------------------------------------------------------ */

int sg2d_term(int px, int py) { const int t = px + py; return t * (t + 1) / 2 + py; }

int sg2d_config_ok(const Savgol2DConfig *c)
{
    if (!c) return 0;
    if (c->half_window_x == 0 || c->half_window_x > SAVGOL2D_MAX_HALF_WINDOW) return 0;
    if (c->half_window_y == 0 || c->half_window_y > SAVGOL2D_MAX_HALF_WINDOW) return 0;
    if (c->poly_order > SAVGOL2D_MAX_POLY_ORDER) return 0;
    if (c->deriv_x + c->deriv_y > c->poly_order) return 0;
    if (!(c->delta_x > 0.0f) || !(c->delta_y > 0.0f)) return 0;
    const int area = (2 * c->half_window_x + 1) * (2 * c->half_window_y + 1);
    return area >= savgol2d_num_terms(c->poly_order);
}

/* Solves the normal equations for the coefficient vector `coef` (length num_terms, double) of the
 * requested pinv row and evaluates the dense kernel W (row-major [2ny+1][2nx+1], float).
 * `coef` is also what the separable decomposition is built from (sg2d_separable_terms). */
int sg2d_weights_fill(const Savgol2DConfig *c, float *W, double *coef)
{
    const int nx = c->half_window_x, ny = c->half_window_y, order = c->poly_order;
    const int ww = 2 * nx + 1, wh = 2 * ny + 1, area = ww * wh;
    const int nt = savgol2d_num_terms(order);
    double *A = (double *)malloc(sizeof(double) * (size_t)area * (size_t)nt);
    double G[SAVGOL2D_MAX_TERMS * SAVGOL2D_MAX_TERMS];
    double fwd[SAVGOL2D_MAX_TERMS];
    if (!A) return -1;

    /* design matrix: one row per window position (y outer, x inner), one column per monomial;
     * powers by repeated multiplication (exact: |x|,|y| <= 16, degree <= 6) */
    double xp[SAVGOL2D_MAX_POLY_ORDER + 1], yp[SAVGOL2D_MAX_POLY_ORDER + 1];
    for (int y = -ny, r = 0; y <= ny; ++y) {
        yp[0] = 1.0;
        for (int e = 1; e <= order; ++e) yp[e] = yp[e - 1] * (double)y;
        for (int x = -nx; x <= nx; ++x, ++r) {
            xp[0] = 1.0;
            for (int e = 1; e <= order; ++e) xp[e] = xp[e - 1] * (double)x;
            for (int px = 0; px <= order; ++px)
                for (int py = 0; px + py <= order; ++py)
                    A[(size_t)r * nt + sg2d_term(px, py)] = xp[px] * yp[py];
        }
    }

    for (int i = 0; i < nt; ++i)
        for (int j = 0; j < nt; ++j) {
            double s = 0.0;
            for (int r = 0; r < area; ++r) s += A[(size_t)r * nt + i] * A[(size_t)r * nt + j];
            G[i * nt + j] = s;
        }

    /* Cholesky G = L L^T in place (lower), then L f = e_target, L^T coef = f */
    for (int i = 0; i < nt; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = G[i * nt + j];
            for (int k = 0; k < j; ++k) s -= G[i * nt + k] * G[j * nt + k];
            if (j < i) G[i * nt + j] = s / G[j * nt + j];
            else if (s > 0.0) G[i * nt + i] = sqrt(s);
            else { free(A); return -1; }
        }
    const int target = sg2d_term(c->deriv_x, c->deriv_y);
    for (int i = 0; i < nt; ++i) {
        double s = (i == target) ? 1.0 : 0.0;
        for (int j = 0; j < i; ++j) s -= G[i * nt + j] * fwd[j];
        fwd[i] = s / G[i * nt + i];
    }
    for (int i = nt - 1; i >= 0; --i) {
        double s = fwd[i];
        for (int j = i + 1; j < nt; ++j) s -= G[j * nt + i] * coef[j];
        coef[i] = s / G[i * nt + i];
    }

    double fx = 1.0, fy = 1.0;
    for (int i = 2; i <= c->deriv_x; ++i) fx *= (double)i;
    for (int i = 2; i <= c->deriv_y; ++i) fy *= (double)i;
    const double dscale = fx * fy;
    for (int r = 0; r < area; ++r) {
        double s = 0.0;
        for (int i = 0; i < nt; ++i) s += A[(size_t)r * nt + i] * coef[i];
        W[r] = (float)(s * dscale);
    }
    for (int i = 0; i < nt; ++i) coef[i] *= dscale;     /* coefficients of the scaled kernel */
    free(A);
    return 0;
}

float sg2d_scale(const Savgol2DConfig *c)
{
    return 1.0f / (powf(c->delta_x, (float)c->deriv_x) * powf(c->delta_y, (float)c->deriv_y));
}
