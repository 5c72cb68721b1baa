"""Does the time of one and the same 1-D batch launch depend on WHERE its buffers are?  Several fresh allocations inside one process,
with dummy allocations of varying size in between (tools, not product)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sg = load_package(); import torch, numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 19
f = sg.Filter(n, 2, 0, 1.0, 1)
ch, length = 4096, 1 << 20
keep = []
for trial in range(8):
    if trial: keep.append(torch.empty((trial * 977 + 13) << 20, dtype=torch.uint8, device="cuda"))     # perturb the allocator
    x = torch.randn((ch, length), device="cuda"); y = torch.empty_like(x)
    for _ in range(2): f.apply_batch(x, y, ch, length)
    torch.cuda.synchronize(); ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f.apply_batch(x, y, ch, length); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"allocation {trial}: x at {x.data_ptr():#x}, y at {y.data_ptr():#x}: median {np.median(ts):.3f} ms", flush=True)
    del x, y
    if trial % 3 == 2: keep.clear(); torch.cuda.empty_cache()
