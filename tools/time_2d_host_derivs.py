"""Host-pointer derivative drop-ins (savgol2d_gradient / _hessian / _laplacian) on one 1024^2 frame against savgol2d_apply (tools)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sg = load_package(); import numpy as np, ctypes as C
L = sg.lib()
rows = cols = 1024
img = np.random.default_rng(0).normal(0, 1, (rows, cols)).astype(np.float32)
o = [np.zeros_like(img) for _ in range(3)]
p = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
def timed(fn, reps=20):
    fn(); fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); rc = fn(); ts.append(time.perf_counter() - t0); assert rc == 0
    return np.median(ts) * 1e3
for n, order in ((7, 3), (16, 6)):
    f = sg.Filter2D(n, n, order)
    for b in (1, 0):
        a = timed(lambda: L.savgol2d_apply(f.ptr, p(img), rows, cols, cols, p(o[0]), cols, b))
        g = timed(lambda: L.savgol2d_gradient(n, n, order, p(img), rows, cols, cols, p(o[0]), p(o[1]), 1.0, 1.0, b))
        h = timed(lambda: L.savgol2d_hessian(n, n, order, p(img), rows, cols, cols, p(o[0]), p(o[1]), p(o[2]), 1.0, 1.0, b))
        l = timed(lambda: L.savgol2d_laplacian(n, n, order, p(img), rows, cols, cols, p(o[0]), 1.0, 1.0, b))
        print(f"n={n} order={order} boundary={b}: apply {a:.3f} ms  gradient {g:.3f}  hessian {h:.3f}  laplacian {l:.3f}", flush=True)
