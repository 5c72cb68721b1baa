"""Host side of the fused stream bank's block-moment tiles (round 5; csrc/sg_stream_moment_fit.cpp, csrc/sg_stream_dma.hip MomGeom / MomTaps): the fit
that decides whether a bank's taps are a polynomial of degree <= 2, and the coefficients the kernel multiplies the block moments with -- rebuilt into
outputs in numpy with the kernel's own grouping (8-tick blocks, direct head / tail taps, moments in the basis 1, t - 3.5, (t - 3.5)^2 - 5.25) and compared
with the double dot product.  No GPU: the fit is host code."""
import ctypes as C

import numpy as np
import pytest

f32, f64 = np.float32, np.float64
OFFSETS = 34


def fma(a, b, c):
    return (np.asarray(a, f64) * np.asarray(b, f64) + np.asarray(c, f64)).astype(f32)


def weights(sg, n, m, d):
    L = sg.lib()
    cfg = sg.SavgolConfig(n, m, d, 1.0, 0)
    f = L.savgol_create(C.byref(cfg))
    assert f
    w = np.array(f.contents.center_weights[:2 * n + 1], f32)
    L.savgol_destroy(f)
    return w


def table(sg, n, w):
    c = np.zeros((3, OFFSETS), f32)
    terms = sg.lib().savgol_hip_stream_moment_table(n, w.ctypes.data_as(C.POINTER(C.c_float)), c.ctypes.data_as(C.POINTER(C.c_float)))
    return terms, c


def expected_terms(m, d):
    """the centre taps of derivative d are an even / odd polynomial in the tap index with d, of degree m or m - 1: 1..3 moments up to degree 2, else 0"""
    deg = m if (m - d) % 2 == 0 else m - 1
    return deg + 1 if deg <= 2 else 0


Q = np.stack([np.ones(8), np.arange(8) - 3.5, (np.arange(8) - 3.5) ** 2 - 5.25]).astype(f32)


def emulate(w, c, terms, x, n, tr=32):
    """one tile of `tr` outputs from rows x[0 : tr + 2n] (x: [rows][streams]) in the kernel's order: rows in arrival order, each feeding its block's
    moments, the outputs it is a direct tap of, and -- when it completes a block -- the block's share of every output that takes the block whole"""
    rows = tr + 2 * n
    acc = [None] * tr
    mom = None
    jf = lambda m: m // 8 + (1 if m % 8 else 0)
    jl = lambda m: (m + 2 * n - 7) // 8
    for r in range(rows):
        j, t = r // 8, r % 8
        xr = x[r]
        if t == 0:
            mom = [xr.copy()] + [(Q[s, t] * xr).astype(f32) for s in range(1, terms)]
        else:
            mom = [(mom[0] + xr).astype(f32)] + [fma(Q[s, t], xr, mom[s]) for s in range(1, terms)]
        for m in range(max(0, r - 2 * n), min(tr - 1, r) + 1):
            whole = jf(m) <= j <= jl(m)
            if not whole:
                k = r - m
                assert k <= 6 or k >= 2 * n - 6
                acc[m] = (w[k] * xr).astype(f32) if acc[m] is None else fma(w[k], xr, acc[m])
        if t == 7:
            for m in range(max(0, r - 2 * n), min(tr - 1, r) + 1):
                if jf(m) <= j <= jl(m):
                    off = 8 * j - m
                    for s in range(terms):
                        acc[m] = (c[s, off] * mom[s]).astype(f32) if acc[m] is None else fma(c[s, off], mom[s], acc[m])
    return np.stack(acc)


CASES = [(n, m, d) for n in (12, 13, 15, 16, 17, 19, 20) for (m, d) in ((0, 0), (1, 0), (2, 0), (3, 0), (1, 1), (2, 1), (2, 2), (3, 2), (3, 1), (4, 0), (4, 1), (5, 2))]


@pytest.mark.parametrize("n,m,d", CASES)
def test_fit_and_block_coefficients(sg, n, m, d):
    w = weights(sg, n, m, d)
    terms, c = table(sg, n, w)
    want = expected_terms(m, d)
    assert terms == want, (n, m, d, terms, want)
    if terms == 0:
        assert not c.any()
        return
    wmax = np.abs(w).max()
    # the coefficients rebuild every tap of every whole block: sum_s c[s][off] q_s(t) = w[off + t] to the fit's tolerance
    for off in range(2 * n - 6):
        rebuilt = (c[:terms, off].astype(f64)[:, None] * Q[:terms].astype(f64)).sum(axis=0)
        assert np.abs(rebuilt - w[off:off + 8]).max() <= 4e-7 * wmax, (n, m, d, off)
    assert not c[:, 2 * n - 6:].any() and not c[terms:].any()
    # outputs in the kernel's arithmetic against the double dot product of the fp32 taps
    rng = np.random.default_rng(n * 100 + m * 10 + d)
    S, tr = 64, 32
    x = (rng.standard_normal((tr + 2 * n, S)) + 0.7 * np.sin(np.arange(tr + 2 * n) * 0.21)[:, None]).astype(f32)
    got = emulate(w, c, terms, x, n, tr)
    ref = np.stack([(w.astype(f64)[:, None] * x[mm:mm + 2 * n + 1].astype(f64)).sum(axis=0) for mm in range(tr)])
    plain = np.stack([np.sum(w[:, None] * x[mm:mm + 2 * n + 1], axis=0, dtype=f32) for mm in range(tr)])
    scale = np.abs(w).astype(f64).sum() * np.abs(x).max()
    err, err_plain = np.abs(got - ref).max() / scale, np.abs(plain - ref).max() / scale
    assert err <= 4e-7, (n, m, d, err, err_plain)


def test_refuses_what_it_cannot_reproduce(sg):
    rng = np.random.default_rng(5)
    n = 16
    w = weights(sg, n, 2, 1)
    noisy = (w + 1e-5 * np.abs(w).max() * rng.standard_normal(w.shape)).astype(f32)
    assert table(sg, n, noisy)[0] == 0
    assert table(sg, 11, weights(sg, 11, 2, 1))[0] == 0                     # below / above the half windows the tiles are built for
    assert table(sg, 21, weights(sg, 21, 2, 1))[0] == 0
    assert sg.lib().savgol_hip_stream_moment_table(16, None, None) == -1
