"""config 1's shape (1 channel x 10^6 samples) back to back: kernel duration and the gap between consecutive kernels (run under rocprofv3 --kernel-trace)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
sg = load_package()
import torch
x = torch.empty((1, 1_000_000), dtype=torch.float32, device="cuda"); sg.synth(x)
y = torch.empty_like(x)
for n, m in ((5, 3), (32, 4)):
    for mode in (0, 1):
        f = sg.Filter(n, m, 0, 1.0, mode)
        for _ in range(100):
            f.apply_batch(x, y, 1, 1_000_000)
        torch.cuda.synchronize()
