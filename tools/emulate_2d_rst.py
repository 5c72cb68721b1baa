"""CPU emulation (numpy, fp32 with one rounding per operation / fused multiply-add) of the vertical B(y) term of the additive 2-D form,
B(j) = b2 j^2, two ways: the shipped folded sum (N fold-adds + N multiply-adds per output) and a SECOND-MOMENT RECURRENCE down the rows
    R' = R - s_out + s_in;  S' = S - R + (N+1) s_out + N s_in;  T' = T - 2 S + R - (N+1)^2 s_out + N^2 s_in      (T = sum j^2 s, S = sum j s, R = sum s)
re-seeded from a direct sum every `seed` rows (a tile's height).  Errors are normwise against the exact double result of the term, scaled by
b2 and divided by max|s| (the 2-D output is ~ |s| for a smoothing kernel), i.e. in the units of the parity bar (1e-6).
    python tools/emulate_2d_rst.py"""
import numpy as np

f32, f64 = np.float32, np.float64


def fma(a, b, c):
    return (f64(a) * f64(b) + f64(c)).astype(f32)


def add(a, b):
    return (f64(a) + f64(b)).astype(f32)


def c2_of(N):
    """the x^2 + y^2 coefficient of the order-2/3 smoothing kernel on a (2N+1)^2 window: w = c0 + c2 (x^2 + y^2)"""
    j = np.arange(-N, N + 1, dtype=f64)
    m0, s2, s4 = (2 * N + 1) ** 2, (2 * N + 1) * np.sum(j ** 2), (2 * N + 1) * np.sum(j ** 4)
    sxy = np.sum(j ** 2) ** 2
    A = np.array([[m0, 2 * s2], [s2, s4 + sxy]])
    c0, c2 = np.linalg.solve(A, np.array([1.0, 0.0]))
    return c0, c2


def run(N, seed, rows=4000, cols=256, dc=0.0, rng=None):
    rng = rng or np.random.default_rng(N)
    y = np.arange(rows)[:, None]
    s = (np.sin(0.03 * y + rng.uniform(0, 6, (1, cols))) + 0.1 * rng.normal(size=(rows, cols)) + dc).astype(f32)
    c0, c2 = c2_of(N)
    b2 = f32(c2)
    j = np.arange(-N, N + 1)
    M = rows - 2 * N
    exact = np.zeros((M, cols), f64)
    for k in range(2 * N + 1):
        exact += f64(b2) * (k - N) ** 2 * s[k:k + M].astype(f64)
    # shipped: folds then multiply-adds, taps b2 j^2 rounded to fp32, k ascending 0..N-1 (tap N is zero)
    taps = (f64(b2) * (np.arange(0, N) - N) ** 2).astype(f32)
    acc = None
    for k in range(N):
        f = add(s[k:k + M], s[2 * N - k:2 * N - k + M])
        acc = (f64(taps[k]) * f64(f)).astype(f32) if acc is None else fma(np.full_like(f, taps[k]), f, acc)
    direct = acc
    # recurrence, re-seeded every `seed` output rows
    rec = np.zeros((M, cols), f32)
    for m0 in range(0, M, seed):
        # direct seed of R, S, T (fp32, folded, exact integer weights)
        R = np.zeros(cols, f32); S = np.zeros(cols, f32); T = np.zeros(cols, f32)
        R = s[m0 + N].copy()
        for k in range(N):
            a, b = s[m0 + k], s[m0 + 2 * N - k]
            e, o = add(a, b), add(b, -a)                    # j = N - k > 0 for row b
            w = f32(N - k)
            R = add(R, e); S = fma(np.full_like(o, w), o, S); T = fma(np.full_like(e, w * w), e, T)
        for m in range(m0, min(m0 + seed, M)):
            rec[m] = (f64(b2) * f64(T)).astype(f32)
            if m + 1 < min(m0 + seed, M):
                so, si = s[m], s[m + 2 * N + 1]
                T = fma(np.full_like(S, f32(-2.0)), S, T); T = add(T, R)
                T = fma(np.full_like(so, f32(-(N + 1) ** 2)), so, T); T = fma(np.full_like(si, f32(N * N)), si, T)
                S = add(S, -R); S = fma(np.full_like(so, f32(N + 1)), so, S); S = fma(np.full_like(si, f32(N)), si, S)
                R = add(add(R, -so), si)
    den = float(np.max(np.abs(s)))
    return float(np.max(np.abs(direct - exact)) / den), float(np.max(np.abs(rec - exact)) / den), float(np.sqrt(np.mean((rec - exact) ** 2)) / den), c2


if __name__ == "__main__":
    for N, seeds in ((3, (20,)), (7, (20, 40)), (10, (10, 20)), (12, (12, 36)), (16, (8, 36, 72))):
        for seed in seeds:
            for dc in (0.0, 100.0):
                d, r, rr, c2 = run(N, seed, dc=dc)
                print(f"N={N:2d} re-seed every {seed:3d} rows, dc={dc:5.0f}: c2={c2:.3e}  direct folded {d:.2e}   recurrence max {r:.2e} rms {rr:.2e}   (x sqrt(2N+1) for the box sum across columns: {r * np.sqrt(2 * N + 1):.2e})", flush=True)
