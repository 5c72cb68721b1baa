import sys, os; sys.path.insert(0, "/root/repo")
os.environ["SAVGOL_HIP_DEBUG"] = "1"
from __graft_entry__ import load_package
sg = load_package(); import torch
x = torch.randn((64, 1 << 16), device="cuda"); y = torch.empty_like(x)
for n in (1, 2, 3, 5, 8, 12, 14, 15, 16):
    sg.Filter(n, 2, 0, 1.0, 1).apply_batch(x, y, 64, 1 << 16)
torch.cuda.synchronize()
