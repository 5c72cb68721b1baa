"""CPU emulation of the two-pass fp32 arithmetic of the fast 2-D kernels (sg_2d_roll.hip) on second-derivative frames: which pass order, which
intermediate precision lands inside max(1e-6, 1.1 x the reference's own error)?  Test infrastructure (imports oracle/); numpy only, no GPU.

    python tools/emulate_2d_passes.py

Every variant is W = sum_t G_t(y) Q_t(x) from an SVD of the reference's fp32 kernel in double (what sg2d_factors_from_kernel does), taps rounded
to fp32, folded symmetric pairs, fused multiply-adds (emulated: exact product in double, one rounding per accumulate).
  vfirst   : vertical pass, result rounded to fp32, horizontal pass            (the shipped order)
  hfirst   : horizontal pass first
  vfirst_hl: vertical pass accumulates in fp32 but keeps an error-free low word for the LDS row (hi + lo), horizontal pass on both
  vfirst_64: vertical pass in double, rounded to hi + lo floats

    python tools/emulate_2d_passes.py --centre
the horizontal-first pass on data with a ramp under the signal (R6.8): plain; centred (taps applied to s_k - c, c = the window's centre sample);
centred + c * sum(rounded taps); centred + c * sigma*, sigma* fitted to the row sums of the reference's dense table (what sg_2d_hf.hip ships).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sgo                                  # noqa: E402
from tests._util import normwise                        # noqa: E402

f32 = np.float32


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def factors(o, n, terms_max=4):
    W = np.asarray(o.W, np.float64).reshape(2 * n + 1, 2 * n + 1)
    U, s, Vt = np.linalg.svd(W)
    r = int(np.sum(s > 1e-7 * s[0]))
    G = [U[:, t] * np.sqrt(s[t]) for t in range(r)]
    Q = [Vt[t] * np.sqrt(s[t]) for t in range(r)]
    return G, Q, r


def conv_fold32(x, taps, axis, n, scale=1.0):
    """folded symmetric/antisymmetric fp32 pass along `axis` of x (valid region only along that axis)"""
    t = (np.asarray(taps, np.float64) * scale).astype(f32)
    sign = 1.0 if abs(taps[0] - taps[2 * n]) <= 1e-6 * np.max(np.abs(taps)) else -1.0
    L = x.shape[axis] - 2 * n
    sl = lambda k: np.take(x, range(k, k + L), axis=axis)
    acc = None
    for k in range(n + 1):
        if k < n:
            f = (sl(k).astype(np.float64) + sign * sl(2 * n - k).astype(np.float64)).astype(f32)
        else:
            f = sl(n) if sign > 0 else np.zeros_like(sl(n))
        tk = np.full(f.shape, t[k], f32)
        acc = (tk.astype(np.float64) * f.astype(np.float64)).astype(f32) if acc is None else fma32(tk, f, acc)
    return acc


def conv_plain32(x, taps, axis, n, scale=1.0, chains=1):
    """unfolded fp32 pass: one fused multiply-add per tap, `chains` interleaved accumulators joined at the end"""
    t = (np.asarray(taps, np.float64) * scale).astype(f32)
    L = x.shape[axis] - 2 * n
    sl = lambda k: np.take(x, range(k, k + L), axis=axis)
    acc = [None] * chains
    for k in range(2 * n + 1):
        tk = np.full(sl(k).shape, t[k], f32)
        c = k % chains
        acc[c] = (tk.astype(np.float64) * sl(k).astype(np.float64)).astype(f32) if acc[c] is None else fma32(tk, sl(k), acc[c])
    r = acc[0]
    for c in range(1, chains):
        if acc[c] is not None:
            r = (r.astype(np.float64) + acc[c].astype(np.float64)).astype(f32)
    return r


def conv64(x, taps, axis, n, scale=1.0):
    t = np.asarray(taps, np.float64).astype(f32).astype(np.float64) * scale
    L = x.shape[axis] - 2 * n
    acc = 0.0
    for k in range(2 * n + 1):
        acc = acc + t[k] * np.take(x, range(k, k + L), axis=axis).astype(np.float64)
    return acc


def run(n, order, dx, dy, seed=0):
    rng = np.random.default_rng(100 + n + seed)
    rows, cols = 200 + n, 317
    yy, xx = np.mgrid[0:rows, 0:cols]
    img = (np.sin(0.05 * xx + 1) * np.cos(0.03 * yy) + 0.001 * yy + rng.normal(0, 0.1, (rows, cols))).astype(f32)
    o = sgo.Filter2D(n, n, order, dx, dy, 0.5, 2.0)
    hi = o.apply_f64acc(img, cols, 0)[n:rows - n, n:cols - n]
    ref = o.apply(img, cols, 0)[n:rows - n, n:cols - n]
    G, Q, r = factors(o, n)
    sc = float(o.scale)
    out = {}
    # vertical first (axis 0 = y with G), then horizontal with Q * scale
    acc_v = 0.0; acc_h = 0.0; acc_hl = 0.0; acc_64 = 0.0; acc_u = 0.0; acc_u2 = 0.0; acc_uu = 0.0; acc_h64 = 0.0; acc_pv_h64 = 0.0
    for t in range(r):
        v = conv_fold32(img, G[t], 0, n)
        acc_v = acc_v + conv_fold32(v, Q[t], 1, n, sc).astype(np.float64) if t else conv_fold32(v, Q[t], 1, n, sc).astype(np.float64)
        h = conv_fold32(img, Q[t], 1, n, sc)
        acc_h = acc_h + conv_fold32(h, G[t], 0, n).astype(np.float64)
        acc_u = acc_u + conv_plain32(v, Q[t], 1, n, sc).astype(np.float64)
        acc_u2 = acc_u2 + conv_plain32(v, Q[t], 1, n, sc, 2).astype(np.float64)
        acc_uu = acc_uu + conv_plain32(conv_plain32(img, G[t], 0, n), Q[t], 1, n, sc, 2).astype(np.float64)
        acc_h64 = acc_h64 + conv64(v, Q[t], 1, n, 1.0) * sc
        acc_pv_h64 = acc_pv_h64 + conv64(conv_plain32(img, G[t], 0, n, 1.0, 2), Q[t], 1, n, 1.0) * sc
        v64 = conv64(img, G[t], 0, n)
        vh = v64.astype(f32); vl = (v64 - vh.astype(np.float64)).astype(f32)
        acc_64 = acc_64 + (conv_fold32(vh, Q[t], 1, n, sc).astype(np.float64) + conv_fold32(vl, Q[t], 1, n, sc).astype(np.float64))
    # single fp32 rounding of multi-term sums (the kernels add terms in fp32)
    out["vfirst"] = normwise(np.asarray(acc_v).astype(f32), hi)
    out["hfirst"] = normwise(np.asarray(acc_h).astype(f32), hi)
    out["v64+hl"] = normwise(np.asarray(acc_64).astype(f32), hi)
    out["v+plainh"] = normwise(np.asarray(acc_u).astype(f32), hi)
    out["v+plainh2"] = normwise(np.asarray(acc_u2).astype(f32), hi)
    out["plainv+plainh2"] = normwise(np.asarray(acc_uu).astype(f32), hi)
    out["v+h64"] = normwise(np.asarray(acc_h64).astype(f32), hi)
    out["plainv2+h64"] = normwise(np.asarray(acc_pv_h64).astype(f32), hi)
    out["ref"] = normwise(ref, hi)
    out["rank"] = r
    return out


f64 = np.float64
E = sys.modules[__name__]


def fold_center(x, taps, axis, n, scale, center, sigma_star=None):
    """folded pass on (x - c), c = the window's own centre sample, + c * sum(taps_fp32)"""
    t=(np.asarray(taps,f64)*scale).astype(f32)
    sigma=float(2.0*np.sum(t[:n].astype(f64))+f64(t[n]))      # the sum of the kernel as the folded pass applies it
    L=x.shape[axis]-2*n
    sl=lambda k: np.take(x, range(k,k+L), axis=axis)
    c=sl(n)
    acc=None
    for k in range(n+1):
        a=(sl(k).astype(f64)-c.astype(f64)).astype(f32)
        if k<n:
            b=(sl(2*n-k).astype(f64)-c.astype(f64)).astype(f32)
            f=(a.astype(f64)+b.astype(f64)).astype(f32)
        else:
            f=a
        tk=np.full(f.shape,t[k],f32)
        acc=(tk.astype(f64)*f.astype(f64)).astype(f32) if acc is None else fma32(tk,f,acc)
    if center=='sigma':
        acc=fma32(np.full(acc.shape,f32(sigma),f32), c, acc)
    if center=='star':
        acc=fma32(np.full(acc.shape,f32(sigma_star),f32), c, acc)
    return acc
def run_centre(n,order,dx,dy,seed=0,ramp=0.002):
    rng=np.random.default_rng(4242+n+seed)
    rows,cols=150+2*n,520
    yy,xx=np.mgrid[0:rows,0:cols]
    img=(np.sin(0.07*xx)*np.cos(0.04*yy)+ramp*xx+rng.normal(0,0.1,(rows,cols))).astype(f32)
    o=sgo.Filter2D(n,n,order,dx,dy,0.5,2.0)
    hi=o.apply_f64acc(img,cols,0)[n:rows-n,n:cols-n]
    ref=o.apply(img,cols,0)[n:rows-n,n:cols-n]
    G,Q,r=factors(o,n); sc=float(o.scale)
    res={}
    W=np.asarray(o.W,f64).reshape(2*n+1,2*n+1)
    rowsum=W.sum(axis=1)*sc
    for name in ('plain','center','sigma','star'):
        acc=0.0
        for t in range(r):
            g32=np.asarray(G[t],f64).astype(f32).astype(f64)
            star=float(np.dot(rowsum,g32)/np.dot(g32,g32))
            if name=='plain': h=conv_fold32(img,Q[t],1,n,sc)
            else: h=fold_center(img,Q[t],1,n,sc,name,star)
            acc=acc+conv_fold32(h,G[t],0,n).astype(f64)
        res[name]=normwise(np.asarray(acc).astype(f32),hi)
    # exact-tap check: how far is the factorised kernel (fp32 taps, exact arithmetic) from the oracle?
    acc=0.0
    for t in range(r):
        acc=acc+conv64(conv64(img,Q[t],1,n,1.0)*1.0,G[t],0,n)*sc if False else acc
    res['ref']=normwise(ref,hi)
    return res


if __name__ == "__main__" and "--centre" in sys.argv:
    for n, order in ((3, 2), (5, 2), (6, 2), (7, 3), (10, 4), (16, 3)):
        for ramp in (0.002, 0.02, 0.2):
            r = run_centre(n, order, 2, 0, ramp=ramp)
            print(f"n={n:2d} order={order} d=(2,0) ramp {ramp:5.3f}/px: " + "  ".join(f"{k} {v:.2e}" for k, v in r.items()), flush=True)
    sys.exit(0)

if __name__ == "__main__":
    for n, order, dx, dy in ((3, 2, 2, 0), (3, 2, 0, 2), (7, 4, 2, 0), (7, 4, 0, 2), (7, 3, 2, 0), (7, 3, 0, 2), (7, 3, 1, 1), (5, 6, 0, 2), (5, 6, 2, 0), (12, 3, 2, 0), (16, 4, 2, 0)):
        res = [run(n, order, dx, dy, s) for s in range(3)]
        keys = ("vfirst", "hfirst", "v64+hl", "v+plainh", "v+plainh2", "plainv+plainh2", "v+h64", "plainv2+h64", "ref")
        print(f"n={n:2d} order={order} d=({dx},{dy}) rank={res[0]['rank']}: " + "  ".join(f"{k} {max(r[k] for r in res):.2e}" for k in keys), flush=True)
