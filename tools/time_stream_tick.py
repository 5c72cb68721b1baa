import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sg = load_package()
import torch
S = 65536
x = torch.randn((256, S), device="cuda"); o = torch.empty(S, device="cuda"); of = torch.empty((33, S), device="cuda")
for n in (1, 4, 8, 16, 24, 32):
    bank = sg.StreamBank(S, n, 2, 1, 1e-3)
    for t in range(2 * n + 2): bank.push(x[t], o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(200): bank.push(x[t % 256], o)
    e1.record(); torch.cuda.synchronize()
    tick = e0.elapsed_time(e1) / 200 * 1e3
    e0.record()
    for t in range(50): bank.flush(of, n)
    e1.record(); torch.cuda.synchronize()
    print(f"n={n:2d}: push {tick:6.2f} us/tick back to back, flush (n rows) {e0.elapsed_time(e1)/50*1e3:7.2f} us")
