#!/usr/bin/env python3
"""A/B builds of libsavgol_hip.so on the 1-D batch kernel over several PLACEMENTS of the two buffers: one big allocation, input at
its start, output displaced by `gap` bytes past the input's end.  A copy on this part moves +-5 % with the distance between what it
reads and what it writes (tools/membench2.hip, 'output displaced'), so a tuning decision needs more than one placement.
   python tools/ab_1d_placements.py libA.so libB.so --n 8 [--f64]"""
import argparse, ctypes as C
import numpy as np, torch
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+"); ap.add_argument("--n", type=int, default=32); ap.add_argument("--m", type=int, default=4)
ap.add_argument("--channels", type=int, default=4096); ap.add_argument("--length", type=int, default=1 << 20)
ap.add_argument("--f64", action="store_true"); ap.add_argument("--deriv", type=int, default=0)
a = ap.parse_args()

class Cfg(C.Structure):
    _fields_ = [("half_window", C.c_uint8), ("poly_order", C.c_uint8), ("derivative", C.c_uint8), ("time_step", C.c_float), ("boundary", C.c_int)]

es = 8 if a.f64 else 4
if a.f64: a.channels //= 2
nbytes = a.channels * a.length * es
gaps = [0, 4096, 1 << 16, (1 << 20) + 8192, (3 << 20) + 256, (16 << 20), (64 << 20) + 12288, (256 << 20) + (1 << 19)]
pool = torch.empty(2 * nbytes + max(gaps) + (1 << 20), dtype=torch.uint8, device="cuda")
base = (pool.data_ptr() + 4095) & ~4095
xin = torch.randn((a.channels, a.length), dtype=torch.float64 if a.f64 else torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
libs = []
for path in a.libs:
    L = C.CDLL(path)
    L.savgol_create.restype = C.c_void_p; L.savgol_create.argtypes = [C.POINTER(Cfg)]
    fn = L.savgol_apply_batch_f64 if a.f64 else L.savgol_apply_batch_f32
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_size_t] * 4 + [C.c_void_p]
    libs.append((path, fn, L.savgol_create(C.byref(Cfg(a.n, a.m, a.deriv, 1.0, 1))), []))
import ctypes
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
for gap in gaps:
    d_in, d_out = base, base + nbytes + gap
    assert hip.hipMemcpyAsync(d_in, xin.data_ptr(), nbytes, 3, st) == 0
    row = f"gap {gap:>10d} B:"
    for path, fn, f, acc in libs:
        run = lambda: fn(f, d_in, d_out, a.channels, a.length, a.length, a.length, st)
        assert run() == 0
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        acc.append(float(np.median(ts))); row += f"  {np.median(ts):7.3f}"
    print(row, flush=True)
for path, fn, f, acc in libs:
    print(f"{path:55s} n={a.n}{' f64' if a.f64 else ''}: mean over placements {np.mean(acc):.3f} ms  min {min(acc):.3f}  max {max(acc):.3f}")
