"""the 2-D additive kernel (config 4's, 64 frames), its n = 2 sibling and the 1-D headline kernel, a few launches each (run under rocprofv3 --pmc)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
sg = load_package()
import torch
x = torch.randn((64, 4096, 4096), device="cuda"); y = torch.empty_like(x)
for n in (7, 2):
    f = sg.Filter2D(n, n, 3 if n > 1 else 2)
    for _ in range(4):
        f.apply_batch(x, y, 4096, 4096, 64, boundary=1, method=2)
torch.cuda.synchronize()
x1 = x.view(1024, -1)[:, :1 << 20].contiguous(); y1 = torch.empty_like(x1)
f1 = sg.Filter(32, 4, 0, 1.0, 1)
for _ in range(4):
    f1.apply_batch(x1, y1, 1024, 1 << 20)
torch.cuda.synchronize()
