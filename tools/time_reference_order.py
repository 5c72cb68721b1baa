import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sg = load_package()
import torch
x = torch.randn((1024, 1<<20), device="cuda"); y = torch.empty_like(x)
L = sg.lib()
for n in (2, 5, 8, 12, 16, 24, 32):
    f = sg.Filter(n, 4, 0, 1.0, 0)
    for opt in (0, 1):
        L.savgol_hip_set_option(sg.SAVGOL_HIP_OPT_REFERENCE_SUMMATION, opt)
        f.apply_batch(x, y, 1024, 1<<20); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f.apply_batch(x, y, 1024, 1<<20); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"n={n} reference_order={opt}: {ms:.3f} ms = {x.numel()/ms/1e6:.1f} Gsamples/s")
    L.savgol_hip_set_option(sg.SAVGOL_HIP_OPT_REFERENCE_SUMMATION, 0)
