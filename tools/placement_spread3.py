"""Which traffic shows the placement spread: reads alone, writes alone, or reads and writes together?  Ten re-allocations of two 16 GiB
tensors in one process; per allocation: torch read-only (sum), write-only (fill), copy, and the 1-D filter both ways (tools)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sg = load_package(); import torch, numpy as np
f = sg.Filter(19, 2, 0, 1.0, 1)
ch, length = 4096, 1 << 20
def t(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
print("trial:  read x  read y | write x write y | copy x->y  y->x | filter x->y  y->x   (ms)")
for trial in range(10):
    x = torch.randn((ch, length), device="cuda"); y = torch.randn((ch, length), device="cuda")
    r = [t(lambda: x.sum()), t(lambda: y.sum()), t(lambda: x.fill_(1.5)), t(lambda: y.fill_(2.5)),
         t(lambda: y.copy_(x)), t(lambda: x.copy_(y)), t(lambda: f.apply_batch(x, y, ch, length)), t(lambda: f.apply_batch(y, x, ch, length))]
    print(f"{trial:5d}: {r[0]:7.3f} {r[1]:7.3f} | {r[2]:7.3f} {r[3]:7.3f} | {r[4]:8.3f} {r[5]:7.3f} | {r[6]:8.3f} {r[7]:7.3f}", flush=True)
    del x, y; torch.cuda.empty_cache()
