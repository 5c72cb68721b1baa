"""Diagnostic: where (rows / columns / frames) the 2-D rolling kernel (method 2) and the tile kernel (method 3) disagree -- found the
missing LDS ordering fence of the branch-free row loop.  python tools/dbg2d.py"""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from __graft_entry__ import load_package
sg = load_package()
for n in (6, 7, 10):
    rng = np.random.default_rng(100 + n)
    images, rows, cols, stride = 2, 300 + n, 617, 624
    yy, xx = np.mgrid[0:rows, 0:cols]
    x = np.zeros((images, rows, stride), np.float32)
    for k in range(images):
        x[k, :, :cols] = (np.sin(0.05 * xx + k) * np.cos(0.03 * yy) + 0.001 * yy + rng.normal(0, 0.1, (rows, cols))).astype(np.float32)
    d = torch.from_numpy(x).cuda()
    f = sg.Filter2D(n, n, 2, 0, 0, 0.5, 2.0)
    for b in range(3):
        got, tile = torch.full_like(d, -5.0), torch.full_like(d, -5.0)
        f.apply_batch(d, got, rows, cols, images, in_stride=stride, out_stride=stride, boundary=b, method=2)
        f.apply_batch(d, tile, rows, cols, images, in_stride=stride, out_stride=stride, boundary=b, method=3)
        g, t = got.cpu().numpy(), tile.cpu().numpy()
        bad = np.argwhere(np.abs(g - t) > 1e-5 * np.abs(t).max())
        print(n, b, "bad:", len(bad), "rows", sorted(set(bad[:, 1]))[:12], "cols", (bad[:, 2].min(), bad[:, 2].max()) if len(bad) else None, "imgs", sorted(set(bad[:,0])))
