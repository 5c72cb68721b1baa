import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sg = load_package()
import ctypes as C
L = sg.lib()
f = sg.Filter(32, 4, 0, 1.0, 0)
for n in (1 << 20, 1 << 24, 1 << 26):
    x = np.random.default_rng(0).normal(0, 1, n).astype(np.float32)
    y = np.empty_like(x)
    px = x.ctypes.data_as(C.POINTER(C.c_float)); py = y.ctypes.data_as(C.POINTER(C.c_float))
    L.savgol_apply(f.ptr, px, py, n)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); rc = L.savgol_apply(f.ptr, px, py, n); t.append(time.perf_counter() - t0)
    best = min(t)
    print(f"savgol_apply host pointers, {n} fp32 samples: {best*1e3:.2f} ms = {n/best/1e6:.0f} Msamples/s = {8*n/best/1e9:.1f} GB/s over PCIe (rc={rc})")
