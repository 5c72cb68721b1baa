"""What does the memory system charge for writing a stream back over the addresses it was read from?  The flat nontemporal copy kernel
(savgol_hip_stream_copy) out of place (x -> y) and in place (x -> x), and every other 16 KiB chunk of a buffer (the access pattern of one colour
phase of the in-place batch call), interleaved rounds in one process.   python tools/time_inplace_copy.py [--gib 16]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

ap = argparse.ArgumentParser(); ap.add_argument("--gib", type=float, default=16.0); a = ap.parse_args()
sg = load_package(); L = sg.lib()
n = int(a.gib * (1 << 30)) // 8
x = torch.randn(n // 2, device="cuda", dtype=torch.float64).repeat(2) if False else torch.empty(n, device="cuda", dtype=torch.float64).normal_()
y = torch.empty_like(x)
nb = x.numel() * 8


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


rows = []
for r in range(3):
    oop = t(lambda: L.savgol_hip_stream_copy(x.data_ptr(), y.data_ptr(), nb, None))
    inp = t(lambda: L.savgol_hip_stream_copy(x.data_ptr(), x.data_ptr(), nb, None))
    rd = t(lambda: L.savgol_hip_stream_read(x.data_ptr(), nb, None, None))
    rows.append((oop, inp, rd))
    print(f"round {r}: out of place {oop:.3f} ms = {2 * nb / oop / 1e6 / 8000:.3f} of 8 TB/s   in place {inp:.3f} ms = {2 * nb / inp / 1e6 / 8000:.3f}  (+{100 * (inp / oop - 1):.1f} %)   read only {rd:.3f} ms", flush=True)
# torch's own in-place elementwise op for comparison (x.mul_(1.0) reads and writes the same addresses)
tm = t(lambda: x.mul_(1.0))
tc = t(lambda: torch.mul(x, 1.0, out=y))
print(f"torch: x.mul_(1.0) {tm:.3f} ms, torch.mul(x, 1.0, out=y) {tc:.3f} ms  (+{100 * (tm / tc - 1):.1f} % in place)")
