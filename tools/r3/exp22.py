"""where do the 17 s per half window of test_fp32_kernels_against_the_reference_s_own_fp32_error go?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
sg = load_package()
from oracle import sgo
import numpy as np, torch
n = 16
x = torch.empty((5, 40000 + 17 * n), dtype=torch.float32, device="cuda"); sg.synth(x, channel0=3 * n)
xh = x.cpu().numpy(); xh64 = xh.astype(np.float64)
T = {"create_o": 0, "f64": 0, "f32": 0, "create_g": 0, "gpu": 0, "norm": 0}
for m in range(0, 7):
    for d in range(0, min(m, 2) + 1):
        for mode, dt in ((0, 1.0), (1, 1.0), (2, 0.5), (3, 1.0)):
            t = time.time(); o = sgo.Filter(n, m, d, dt, mode); T["create_o"] += time.time() - t
            t = time.time(); r64 = o.apply_f64(xh64); T["f64"] += time.time() - t
            t = time.time(); r32 = o.apply(xh); T["f32"] += time.time() - t
            t = time.time(); f = sg.Filter(n, m, d, dt, mode); T["create_g"] += time.time() - t
            t = time.time(); got = f.apply_tensor(x).cpu().numpy(); T["gpu"] += time.time() - t
            t = time.time(); e = np.max(np.abs(got - r64)) / np.max(np.abs(r64)); T["norm"] += time.time() - t
print({k: round(v, 2) for k, v in T.items()})
