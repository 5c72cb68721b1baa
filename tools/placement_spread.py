"""Does the time of one and the same 1-D batch launch depend on WHERE its two 16 GiB buffers are?  (tools, not product)

    python tools/placement_spread.py --mode perturb [--n 19]   fresh allocations with dummy allocations of varying size in between
    python tools/placement_spread.py --mode swap               ten re-allocations, each also with the two buffers' roles swapped; then
                                                               one pool holding both buffers at three gaps
    python tools/placement_spread.py --mode traffic            per allocation: read-only, write-only, copy, and the filter both ways
    python tools/placement_spread.py --mode tiles [--n 8]      narrow vs wide tile on the same allocations (per-call flags)

profiles/r02_placement_spread.txt holds the round-2 outputs of the first three: the same launch runs 5.18-5.75 ms, reads alone
and writes alone do not vary, reads AND writes together do (and differently for x->y and y->x of one pair)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sg = load_package()
import numpy as np, torch

ap = argparse.ArgumentParser()
ap.add_argument("--mode", choices=["perturb", "swap", "traffic", "tiles"], default="perturb")
ap.add_argument("--n", type=int, default=19)
a = ap.parse_args()
ch, length = 4096, 1 << 20
f = sg.Filter(a.n, 2, 0, 1.0, 1)


def t(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def filt(x, y, flags=None):
    return lambda: f.apply_batch(x, y, ch, length, flags=flags)


if a.mode == "perturb":
    keep = []
    for trial in range(8):
        if trial: keep.append(torch.empty((trial * 977 + 13) << 20, dtype=torch.uint8, device="cuda"))     # perturb the allocator
        x = torch.randn((ch, length), device="cuda"); y = torch.empty_like(x)
        print(f"allocation {trial}: x at {x.data_ptr():#x}, y at {y.data_ptr():#x}: median {t(filt(x, y), 7):.3f} ms", flush=True)
        del x, y
        if trial % 3 == 2: keep.clear(); torch.cuda.empty_cache()
elif a.mode == "swap":
    for trial in range(10):
        x = torch.randn((ch, length), device="cuda"); y = torch.empty_like(x)
        print(f"trial {trial}: x {x.data_ptr():#x} y {y.data_ptr():#x}: {t(filt(x, y)):.3f} ms, swapped {t(filt(y, x)):.3f} ms", flush=True)
        del x, y; torch.cuda.empty_cache()
    pool = torch.empty(2 * ch * length * 4 + (64 << 20), dtype=torch.uint8, device="cuda")
    x = pool[: ch * length * 4].view(torch.float32).view(ch, length); x.normal_()
    for off in (0, 2 << 20, 32 << 20):
        y = pool[ch * length * 4 + off: 2 * ch * length * 4 + off].view(torch.float32).view(ch, length)
        print(f"pool, gap {off >> 20} MiB: {t(filt(x, y)):.3f} ms, swapped {t(filt(y, x)):.3f} ms", flush=True)
elif a.mode == "traffic":
    print("trial:  read x  read y | write x write y | copy x->y  y->x | filter x->y  y->x   (ms)")
    for trial in range(10):
        x = torch.randn((ch, length), device="cuda"); y = torch.randn((ch, length), device="cuda")
        r = [t(lambda: x.sum()), t(lambda: y.sum()), t(lambda: x.fill_(1.5)), t(lambda: y.fill_(2.5)),
             t(lambda: y.copy_(x)), t(lambda: x.copy_(y)), t(filt(x, y)), t(filt(y, x))]
        print(f"{trial:5d}: {r[0]:7.3f} {r[1]:7.3f} | {r[2]:7.3f} {r[3]:7.3f} | {r[4]:8.3f} {r[5]:7.3f} | {r[6]:8.3f} {r[7]:7.3f}", flush=True)
        del x, y; torch.cuda.empty_cache()
else:
    print(f"n={a.n}: trial: narrow x->y  y->x | wide x->y  y->x  (ms)")
    for trial in range(10):
        x = torch.randn((ch, length), device="cuda"); y = torch.randn((ch, length), device="cuda")
        N, W = sg.SAVGOL_BATCH_TILE_NARROW, sg.SAVGOL_BATCH_TILE_WIDE
        print(f"{trial:3d}: {t(filt(x, y, N)):8.3f} {t(filt(y, x, N)):7.3f} | {t(filt(x, y, W)):8.3f} {t(filt(y, x, W)):7.3f}", flush=True)
        del x, y; torch.cuda.empty_cache()
