"""what centring costs the 1-D fp32 derivative filters (R6.16): ms per launch, d = 0 against d = 1, 2 at the headline's shape (2048 ch x 2^20)
   python tools/time_1d_derivative.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sg = load_package()
ch, L = 2048, 1 << 20
x = torch.empty((ch, L), dtype=torch.float32, device="cuda"); sg.synth(x)
y = torch.empty_like(x)
ev = lambda: torch.cuda.Event(enable_timing=True)
for n in (8, 16, 25, 32):
    row = []
    for d in (0, 1, 2):
        f = sg.Filter(n, 4, d, 1.0, 0)
        for _ in range(3): f.apply_batch(x, y, ch, L)
        torch.cuda.synchronize(); ts = []
        for _ in range(5):
            e0, e1 = ev(), ev(); e0.record()
            for _ in range(4): f.apply_batch(x, y, ch, L)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 4)
        ms = float(np.median(ts)); row.append(f"d={d}: {ms:.3f} ms ({8.0 * ch * L / (ms * 1e-3) / 8e12:.3f})")
    print(f"n={n} m=4: " + "   ".join(row), flush=True)
