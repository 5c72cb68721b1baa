#!/usr/bin/env python3
"""Small driver for profiling the 1-D batch kernels under rocprofv3:
    python tools/run_1d.py --n 32 --dtype f32 --channels 4096 --length 1048576 --iters 5 --mode 1
Prints the per-launch time measured with HIP events (same stream as the launches)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--m", type=int, default=4)
ap.add_argument("--d", type=int, default=0)
ap.add_argument("--mode", type=int, default=1)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--channels", type=int, default=4096)
ap.add_argument("--length", type=int, default=1 << 20)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--fill", default="synth", help="synth | zero | f32 (fp32-representable values)")
ap.add_argument("--copy", action="store_true", help="also time a device-to-device copy of the same bytes")
a = ap.parse_args()

sg = load_package()
tdt = torch.float32 if a.dtype == "f32" else torch.float64
x = torch.empty((a.channels, a.length), dtype=tdt, device="cuda")
y = torch.empty_like(x)
sg.synth(x)
if a.fill == 'zero':
    x.zero_()
elif a.fill == 'f32':
    x.copy_(x.float().to(tdt))
f = sg.Filter(a.n, a.m, a.d, 1.0, a.mode)
f.apply_batch(x, y, a.channels, a.length, dtype=a.dtype)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
for e0, e1 in ev:
    e0.record(); f.apply_batch(x, y, a.channels, a.length, dtype=a.dtype); e1.record()
torch.cuda.synchronize()
ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
byt = 2.0 * x.numel() * x.element_size()
print(f"n={a.n} {a.dtype} mode={a.mode} {a.channels}x{a.length}: median {ms[len(ms)//2]:.3f} ms  min {ms[0]:.3f} ms  "
      f"-> {byt/ms[len(ms)//2]/1e6:.1f} GB/s algorithmic, {x.numel()/ms[len(ms)//2]/1e3:.1f} Msamples/s")
if a.copy:
    for e0, e1 in ev:
        e0.record(); y.copy_(x); e1.record()
    torch.cuda.synchronize()
    ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    print(f"torch copy_ same bytes: median {ms[len(ms)//2]:.3f} ms -> {byt/ms[len(ms)//2]/1e6:.1f} GB/s")
