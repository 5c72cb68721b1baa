"""One in-place fp64 chunk of config 5 (1024 x 2^22, n = 32, d = 2) a few times, for rocprofv3 --kernel-trace: which launch of the call costs what.
   python tools/run_inplace.py [--tol 1e-6] [--channels 1024]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
ap = argparse.ArgumentParser(); ap.add_argument("--tol", type=float, default=1e-6); ap.add_argument("--channels", type=int, default=1024); ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
sg = load_package()
ch, L = a.channels, 1 << 22
x = torch.empty((ch, L), dtype=torch.float64, device="cuda"); sg.synth(x)
y = torch.empty_like(x)
f = sg.Filter(32, 4, 2, 1.0, 0)
kw = {"rel_tol": a.tol} if a.tol > 0 else {"flags": 0}
for i in range(a.reps):
    f.apply_batch(x, y, ch, L, dtype="f64", **kw)          # out of place
torch.cuda.synchronize()
for i in range(a.reps):
    y.copy_(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f.apply_batch(y, y, ch, L, dtype="f64", **kw); e1.record(); torch.cuda.synchronize()
    print("in place", e0.elapsed_time(e1), "ms")
