"""Workload for PMC passes on the 2-D rolling kernel at a small half window (n=2, order 2), 64 frames of 4096^2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sg = load_package()
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
x = torch.randn((64, 4096, 4096), device="cuda"); y = torch.empty_like(x)
f = sg.Filter2D(n, n, min(3, 2 * n))
for _ in range(4):
    f.apply_batch(x, y, 4096, 4096, 64, boundary=1, method=2)
torch.cuda.synchronize()
