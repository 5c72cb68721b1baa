import os, sys, time
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
sg = load_package()
import numpy as np
x = np.random.default_rng(1).normal(0, 1, 300).astype(np.float32)
s = sg.Stream(16, 2, 1, 1e-3)
for i, v in enumerate(x[:40]):
    t0 = time.perf_counter(); r = s.push(float(v)); dt = time.perf_counter() - t0
    if i >= 30: print(i, r, f"{dt*1e6:.1f} us", sg.last_error())
