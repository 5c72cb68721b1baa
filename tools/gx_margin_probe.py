"""How much of the fused gradient's gx error (vertical-first: the x-derivative pass runs LAST) is driven by an offset / ramp under the signal?
   python tools/gx_margin_probe.py      prints value / bar of gx and gy for n = 2, 3, 4, 5, 7, order 3, VALID, with the ramp scaled 0 / 1 / 10 x"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
from oracle import sgo
from tests._util import normwise, fp32_bar
sg = load_package(); L = sg.lib()
rng = np.random.default_rng(808 + 3)
rows, cols = 126, 520
yy, xx = np.mgrid[0:rows, 0:cols]
base = np.sin(0.07 * xx + 1) * np.cos(0.04 * yy) + rng.normal(0, 0.1, (rows, cols))
for n in (2, 3, 4, 5, 7):
    for ramp in (0.0, 1.0, 10.0, 100.0):
        x = (base + ramp * 0.002 * xx).astype(np.float32)
        d = torch.from_numpy(x).cuda()
        gx, gy = torch.zeros_like(d), torch.zeros_like(d)
        assert L.savgol2d_gradient_batch_f32(n, n, 3, d.data_ptr(), rows, cols, cols, rows * cols, gx.data_ptr(), gy.data_ptr(), cols, rows * cols, 1, 0.5, 2.0, 0, None) == 0
        out = []
        for name, got, (dx, dy) in (("gx", gx, (1, 0)), ("gy", gy, (0, 1))):
            o = sgo.Filter2D(n, n, 3, dx, dy, 0.5, 2.0)
            hi = o.apply_f64acc(x, cols, 0)[n:rows - n, n:cols - n]
            ref = o.apply(x, cols, 0)[n:rows - n, n:cols - n]
            g = got.cpu().numpy()[n:rows - n, n:cols - n]
            e, er = normwise(g, hi), normwise(ref, hi)
            out.append(f"{name}: ours {e:.2e} ref {er:.2e} ours/bar {e / fp32_bar(er):.2f}")
        print(f"n={n} ramp x{ramp:<5}: " + "   ".join(out), flush=True)
