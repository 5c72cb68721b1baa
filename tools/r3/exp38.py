"""host-pointer savgol_apply (n = 5, POLYNOMIAL) on 2^19 ... 2^24 samples: the plain path against the pipelined one at several thresholds (env per process)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
sg = load_package()
import numpy as np
f = sg.Filter(5, 3, 0, 1.0, 0)
out = []
for lg in (19, 20, 21, 22, 23, 24):
    L = 1 << lg
    x = np.random.default_rng(0).normal(0, 1, L).astype(np.float32); y = np.zeros_like(x)
    for _ in range(3): f.apply(x, out=y)
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); f.apply(x, out=y); ts.append(time.perf_counter() - t0)
    out.append(f"2^{lg}: {np.median(ts) * 1e6:8.1f} us")
print("  ".join(out))
