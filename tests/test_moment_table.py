"""Host logic of the fp32 block-moment kernel (csrc/sg_k1d_momenth.hpp, half windows 20..32; host fit: csrc/sg_k1d_moment_fit.cpp), no GPU needed:
the constant table the kernel reads is rebuilt into outputs with the kernel's own arithmetic (fp32 FMA chains in the kernel's order, emulated in
numpy) and compared with the fp64 oracle.  The reference loop this path replaces: savgol_apply centre loop,
/root/reference/src/savgolFilter.c:763-766 with convolve_ilp :547-580.  (Round 2's whole-lane form and its table went in round 6.)"""
import ctypes as C

import numpy as np
import pytest

from oracle import sgo

FLOATS = 400
f32, f64 = np.float32, np.float64


def fma(a, b, c):                       # one rounding, like v_fma_f32 / v_pk_fma_f32
    return (f64(a) * f64(b) + f64(c)).astype(f32)


# ---- the half-lane form (csrc/sg_k1d_momenth.hpp; table layout in csrc/sg_k1d_host.hpp) ----
H_OFF_W, H_OFF_PHI, H_OFF_C = 0, 132, 276


def geometry_h(n):
    off = (n + 3) // 4 * 4 - n
    return off, (15 + off + 1) // 2 * 2, (off + 2 * n + 1) // 2 * 2


def table_h(sg, n, m, d):
    L = sg.lib()
    cfg = sg.SavgolConfig(n, m, d, 1.0, 0)
    f = L.savgol_create(C.byref(cfg))
    assert f
    tab = np.zeros(FLOATS, f32)
    terms = L.savgol_hip_momenth_table(f, tab.ctypes.data_as(C.POINTER(C.c_float)))
    w = np.array(f.contents.center_weights[:2 * n + 1], f32)
    L.savgol_destroy(f)
    return terms, tab, w


def emulate_h(tab, terms, x, n):
    """every 16-output group of x in the kernel's own arithmetic: x-stationary head and tail on three round-robin chains (tap pair (w[k], w[k-1])
    from the table), chains joined, block samples paired front to back into two partial moments each, the block's share added last"""
    off, lo, hi = geometry_h(n)
    bk, steps = hi - lo, (hi - lo) // 4
    wp = tab[H_OFF_W:H_OFF_W + 2 * (2 * n + 2)].reshape(-1, 2)              # wp[k] = (w[k], w[k-1])
    groups = (len(x) - 2 * n) // 16
    xp = np.concatenate([np.zeros(off, f32), x, np.zeros(8, f32)])
    X = np.stack([xp[16 * g:16 * g + 16 + 2 * n + off + 4] for g in range(groups)])
    A = np.zeros((3, 16, groups), f32)
    for i in list(range(off, lo)) + list(range(hi, off + 2 * n + 16)):
        for j in range(8):
            k = i - off - 2 * j
            if (i < lo and 0 <= k <= lo - 1 - off) or (i >= hi and hi - off - 14 <= k <= 2 * n + 1):
                A[k % 3, 2 * j] = fma(wp[k, 0], X[:, i], A[k % 3, 2 * j])
                A[k % 3, 2 * j + 1] = fma(wp[k, 1], X[:, i], A[k % 3, 2 * j + 1])
    a0 = (A[0] + A[1]).astype(f32) + A[2]
    M = np.zeros((terms, 2, groups), f32)
    for u in range(steps):
        f0, f1 = X[:, lo + 2 * u], X[:, lo + 2 * u + 1]
        b0, b1 = X[:, hi - 1 - 2 * u], X[:, hi - 2 - 2 * u]
        e = np.stack([f0 + b0, f1 + b1]).astype(f32)
        o = np.stack([f0 - b0, f1 - b1]).astype(f32)
        M[0] = e if u == 0 else (M[0] + e).astype(f32)
        for s in range(1, terms):
            ph = tab[H_OFF_PHI + (u * 6 + s - 1) * 2:H_OFF_PHI + (u * 6 + s - 1) * 2 + 2]
            v = o if s & 1 else e
            M[s] = np.stack([fma(ph[0], v[0], M[s, 0] if u else np.zeros(groups, f32)), fma(ph[1], v[1], M[s, 1] if u else np.zeros(groups, f32))])
    mu = (M[:, 0] + M[:, 1]).astype(f32)
    B = np.zeros((16, groups), f32)
    for s in range(terms - 1, -1, -1):
        c = tab[H_OFF_C + s * 16:H_OFF_C + s * 16 + 16]
        for r in range(16):
            B[r] = fma(c[r], mu[s], B[r])
    return (a0 + B).astype(f32).T.reshape(-1)


@pytest.mark.parametrize("n", [32, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20])
@pytest.mark.parametrize("m,d,terms_expected,tol", [(4, 0, 5, 1e-6), (2, 0, 3, 1e-6), (3, 0, 3, 1e-6), (6, 0, 7, 1e-6), (4, 1, 5, 2e-6), (5, 1, 7, 2e-6)])
def test_half_lane_table_reproduces_the_filter(sg, n, m, d, terms_expected, tol):
    """the table of round 5's fp32 kernel (savgol_hip_momenth_table), rebuilt into outputs with the kernel's own arithmetic in numpy and compared with
    the double-precision convolution of the same fp32 taps; geometry and layout as csrc/sg_k1d_host.hpp states them"""
    terms, tab, w = table_h(sg, n, m, d)
    if terms == 0 and n < 24:
        pytest.skip("the fit does not reproduce this table to 3e-7 of its largest tap (n = 21, m = 5, d = 1): the filter keeps the plain sum")
    assert terms == terms_expected
    off, lo, hi = geometry_h(n)
    assert lo % 2 == 0 and hi % 2 == 0 and (hi - lo) % 4 == 0 and lo >= 15 + off and hi <= off + 2 * n + 1 and 20 <= hi - lo <= 48
    wp = tab[H_OFF_W:H_OFF_W + 2 * (2 * n + 2)].reshape(-1, 2)
    assert np.array_equal(wp[:2 * n + 1, 0], w) and wp[2 * n + 1, 0] == 0 and wp[0, 1] == 0 and np.array_equal(wp[1:, 1], w)
    x = sgo.synth_f32(5, 1, 16 * 128 + 2 * n)[0]
    got = emulate_h(tab, terms, x, n)
    ref = np.convolve(x.astype(f64), w[::-1].astype(f64), "valid")[:len(got)]
    err = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
    assert err < tol, f"n={n} m={m} d={d}: normwise error {err:.3e}"


def test_paths_that_keep_the_plain_sum(sg):
    assert table_h(sg, 32, 8, 0)[0] == 0            # poly_order > 6
    assert table_h(sg, 16, 4, 0)[0] == 0 and table_h(sg, 19, 4, 0)[0] == 0      # half windows below MOMENTH_MIN_N = 20
    # a hand-edited table is not a polynomial: the fit must refuse it
    L = sg.lib()
    cfg = sg.SavgolConfig(32, 4, 0, 1.0, 0)
    f = L.savgol_create(C.byref(cfg))
    f.contents.center_weights[10] += 1e-4
    tab = np.zeros(FLOATS, f32)
    assert L.savgol_hip_momenth_table(f, tab.ctypes.data_as(C.POINTER(C.c_float))) == 0
    L.savgol_destroy(f)
    assert L.savgol_hip_momenth_table(None, tab.ctypes.data_as(C.POINTER(C.c_float))) == -1


def test_every_moment_table_is_compatible_with_the_1e6_bar(sg):
    """the fit accepts a polynomial that reproduces each fp32 tap to 3e-7 of the largest one; the taps of a whole block could
    add up.  So bound what actually reaches an output, for EVERY filter savgol_create builds at half windows 20..32 (every poly_order, every
    derivative): the kernel replaces the taps on a group's common block by sum_s c_s(r) phi_s(t), so output r carries the error
    E_r = sum_t |sum_s c_s(r) phi_s(t) - w[lo + t - r - off]|  per unit of input amplitude, against sum |w| -- the largest output a unit-amplitude
    input can produce.  E_r / sum|w| must leave the fp32 rounding of the remaining sum inside 1e-6: the worst table measures well under the 1.5e-7
    asserted here (printed with -s)."""
    worst = (0.0, None)
    covered = 0
    for n in range(20, 33):
        off, lo, hi = geometry_h(n)
        bk = hi - lo
        for m in range(0, 11):
            for d in range(0, min(m, 4) + 1):
                if 2 * n + m + 1 >= 76:
                    continue
                terms, tab, w = table_h(sg, n, m, d)
                if terms == 0:
                    continue
                covered += 1
                phi = np.ones((terms, bk), f64)
                for s in range(1, terms):
                    half = np.array([tab[H_OFF_PHI + ((t // 2) * 6 + s - 1) * 2 + (t & 1)] for t in range(bk // 2)], f64)
                    phi[s, :bk // 2] = half
                    phi[s, bk // 2:] = half[::-1] * (-1.0 if s & 1 else 1.0)
                c = np.stack([tab[H_OFF_C + s * 16:H_OFF_C + (s + 1) * 16].astype(f64) for s in range(terms)])       # [s][r]
                w_eff = c.T @ phi                                                                                   # [r][t]
                sum_abs = np.sum(np.abs(w.astype(f64)))
                for r in range(16):
                    true = w[lo - r - off:hi - r - off].astype(f64)
                    e = float(np.sum(np.abs(w_eff[r] - true)) / sum_abs)
                    if e > worst[0]:
                        worst = (e, (n, m, d, r, terms))
    print(f"moment tables checked: {covered}; worst block error / sum|w| = {worst[0]:.3e} at (n, m, d, r, terms) = {worst[1]}")
    assert covered >= 13 * 18
    assert worst[0] <= 1.5e-7, worst
