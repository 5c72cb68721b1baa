import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from __graft_entry__ import load_package
sg = load_package()
from oracle import sgo
from tests._util import normwise
for n in (9, 11, 13, 14, 15, 16):
    rng = np.random.default_rng(100 + n)
    images, rows, cols, stride = 2, 300 + n, 617, 624
    yy, xx = np.mgrid[0:rows, 0:cols]
    x = np.zeros((images, rows, stride), np.float32)
    for k in range(images):
        x[k, :, :cols] = (np.sin(0.05 * xx + k) * np.cos(0.03 * yy) + 0.001 * yy + rng.normal(0, 0.1, (rows, cols))).astype(np.float32)
    d = torch.from_numpy(x).cuda()
    for order, dx, dy in ((3, 1, 0), (4, 0, 0), (6, 0, 2)):
        f = sg.Filter2D(n, n, order, dx, dy, 0.5, 2.0); o = sgo.Filter2D(n, n, order, dx, dy, 0.5, 2.0)
        b = 1
        hi = o.apply_f64acc(x[1], cols, b)
        line = f"n={n} order={order} d=({dx},{dy}):"
        for m in (2, 3):
            got = torch.full_like(d, -5.0)
            f.apply_batch(d, got, rows, cols, images, in_stride=stride, out_stride=stride, boundary=b, method=m)
            g = got.cpu().numpy()
            line += f"  method {m}: {normwise(g[1][:, :cols], hi[:, :cols]):.2e}"
        # the reference's own fp32 dense sum against the same double oracle
        ref32 = o.apply(x[1], cols, b)
        line += f"  reference fp32 order: {normwise(ref32[:, :cols], hi[:, :cols]):.2e}"
        print(line, flush=True)
