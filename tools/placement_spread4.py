"""Do the narrow and the wide tile of one kernel prefer different placements of the same two buffers?  (SAVGOL_HIP_OPT_TILE_WIDTH; tools)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sg = load_package(); import torch, numpy as np
L = sg.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
f = sg.Filter(n, 2, 0, 1.0, 1)
ch, length = 4096, 1 << 20
def t(x, y, width):
    L.savgol_hip_set_option(sg.SAVGOL_HIP_OPT_TILE_WIDTH, width)
    for _ in range(2): f.apply_batch(x, y, ch, length)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f.apply_batch(x, y, ch, length); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
print(f"n={n}: trial: narrow x->y  y->x | wide x->y  y->x  (ms)")
for trial in range(10):
    x = torch.randn((ch, length), device="cuda"); y = torch.randn((ch, length), device="cuda")
    print(f"{trial:3d}: {t(x, y, 1):8.3f} {t(y, x, 1):7.3f} | {t(x, y, 2):8.3f} {t(y, x, 2):7.3f}", flush=True)
    del x, y; torch.cuda.empty_cache()
L.savgol_hip_set_option(sg.SAVGOL_HIP_OPT_TILE_WIDTH, 0)
