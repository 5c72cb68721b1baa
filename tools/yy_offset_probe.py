"""single-filter 2-D second derivatives on frames with a constant offset: d = (0,2) runs vertical-first on RAW rows (its x twin (2,0) runs on centred samples: sg_2d_hf.hip)
   python tools/yy_offset_probe.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
from oracle import sgo
from tests._util import normwise, fp32_bar
sg = load_package()
rng = np.random.default_rng(9)
rows, cols = 126, 520
yy, xx = np.mgrid[0:rows, 0:cols]
base = np.sin(0.07 * xx + 1) * np.cos(0.04 * yy) + rng.normal(0, 0.1, (rows, cols))
for n in (3, 7, 12):
    for (dx, dy) in ((0, 2), (2, 0), (0, 1), (1, 0)):
        row = []
        for off in (0.0, 10.0, 100.0):
            x = (base + off).astype(np.float32)
            d = torch.from_numpy(x).cuda(); out = torch.zeros_like(d)
            f = sg.Filter2D(n, n, 3, dx, dy, 0.5, 2.0); o = sgo.Filter2D(n, n, 3, dx, dy, 0.5, 2.0)
            f.apply_batch(d, out, rows, cols, 1, boundary=0, method=2)
            hi = o.apply_f64acc(x, cols, 0)[n:rows - n, n:cols - n]; ref = o.apply(x, cols, 0)[n:rows - n, n:cols - n]
            g = out.cpu().numpy()[n:rows - n, n:cols - n]
            e, er = normwise(g, hi), normwise(ref, hi)
            row.append(f"off {off:g}: ours {e:.1e} ref {er:.1e} ({e / fp32_bar(er):.2f})")
        print(f"n={n} d=({dx},{dy}): " + "   ".join(row), flush=True)
