"""config 1 through the host-pointer drop-in call, 30 calls (run under rocprofv3 --kernel-trace --memory-copy-trace): where do the ~0.2 ms go?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
sg = load_package()
import numpy as np
x = np.random.default_rng(0).normal(0, 1, 1_000_000).astype(np.float32); y = np.zeros_like(x)
f = sg.Filter(5, 3, 0, 1.0, 0)
for _ in range(5): f.apply(x, out=y)
ts = []
for _ in range(30):
    t0 = time.perf_counter(); f.apply(x, out=y); ts.append(time.perf_counter() - t0)
print("median host call %.1f us" % (np.median(ts) * 1e6))
