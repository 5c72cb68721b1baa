"""Wall time of the host-pointer drop-in call (savgol_apply) across signal lengths, same buffers reused (pages already touched)."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sg = load_package()
import ctypes as C
L = sg.lib()
f = sg.Filter(5, 3, 0, 1.0, 0)
for n in (360, 1024, 4096, 1 << 16, 1 << 18, 1000000, 1 << 22, (1 << 23) - 8, 1 << 23, 1 << 24):
    x = np.random.default_rng(0).normal(0, 1, n).astype(np.float32)
    y = np.zeros_like(x)
    px = x.ctypes.data_as(C.POINTER(C.c_float)); py = y.ctypes.data_as(C.POINTER(C.c_float))
    for _ in range(3): L.savgol_apply(f.ptr, px, py, n)
    t = []
    for _ in range(9):
        t0 = time.perf_counter(); rc = L.savgol_apply(f.ptr, px, py, n); t.append(time.perf_counter() - t0)
    print(f"savgol_apply host pointers, {n:9d} fp32 samples: median {np.median(t)*1e3:8.3f} ms  min {min(t)*1e3:8.3f} ms = {n/min(t)/1e6:8.1f} Msamples/s (rc={rc})", flush=True)
