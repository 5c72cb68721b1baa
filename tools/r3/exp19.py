"""one order-6 call at n = 9 and n = 16 (two rolling passes) under rocprofv3 --kernel-trace: how long does each pass take?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
sg = load_package()
import torch
images, size = 16, 4096
x = torch.randn((images, size, size), device="cuda")
y = torch.empty_like(x)
for n, order in ((9, 3), (9, 6), (16, 3), (16, 6), (13, 4)):
    f = sg.Filter2D(n, n, order)
    for _ in range(3):
        f.apply_batch(x, y, size, size, images, boundary=1, method=2)
    torch.cuda.synchronize()
