import sys, os; sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
sg = load_package(); import torch, numpy as np
f = sg.Filter(19, 2, 0, 1.0, 1)
ch, length = 4096, 1 << 20
def t(x, y):
    for _ in range(2): f.apply_batch(x, y, ch, length)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f.apply_batch(x, y, ch, length); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for trial in range(10):
    x = torch.randn((ch, length), device="cuda"); y = torch.empty_like(x)
    a = t(x, y); b = t(y, x)          # same two buffers, roles swapped
    print(f"trial {trial}: x {x.data_ptr():#x} y {y.data_ptr():#x}: {a:.3f} ms, swapped {b:.3f} ms", flush=True)
    del x, y; torch.cuda.empty_cache()
# one pool, x and y inside
pool = torch.empty(2 * ch * length * 4 + (64 << 20), dtype=torch.uint8, device="cuda")
x = pool[: ch * length * 4].view(torch.float32).view(ch, length); x.normal_()
for off in (0, 2 << 20, 32 << 20):
    y = pool[ch * length * 4 + off: 2 * ch * length * 4 + off].view(torch.float32).view(ch, length)
    print(f"pool, gap {off >> 20} MiB: {t(x, y):.3f} ms, swapped {t(y, x):.3f} ms", flush=True)
