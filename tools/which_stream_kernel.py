"""which kernel a fused bank's block push runs (for rocprofv3 --kernel-trace): python tools/which_stream_kernel.py n m d"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sg = load_package()
n, m, d = (int(v) for v in sys.argv[1:4])
S, T = 256, 1024
x = torch.randn((T, S), device="cuda") + 1000.0
out = torch.zeros_like(x)
bank = sg.StreamBank(S, n, m, d, 1.0, fma=True)
print("rows", bank.push_block(x, T, out)); torch.cuda.synchronize()
