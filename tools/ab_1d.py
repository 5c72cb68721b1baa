#!/usr/bin/env python3
"""A/B two builds of libsavgol_hip.so on the 1-D batch kernel in ONE process, interleaved rounds (rule 24 of the
CDNA guide: never rank builds across processes/devices).   python tools/ab_1d.py libA.so libB.so [--n 32]"""
import argparse
import ctypes as C
import sys

import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--m", type=int, default=4)
ap.add_argument("--channels", type=int, default=4096)
ap.add_argument("--length", type=int, default=1 << 20)
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--f64", action="store_true", help="savgol_apply_batch_f64 on float64 data")
ap.add_argument("--deriv", type=int, default=0)
ap.add_argument("--zeros", action="store_true", help="all-zero input (data-dependent power)")
a = ap.parse_args()


class Cfg(C.Structure):
    _fields_ = [("half_window", C.c_uint8), ("poly_order", C.c_uint8), ("derivative", C.c_uint8), ("time_step", C.c_float), ("boundary", C.c_int)]


x = (torch.zeros if a.zeros else torch.randn)((a.channels, a.length), dtype=torch.float64 if a.f64 else torch.float32, device="cuda")
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
libs = []
for spec in a.libs:
    # path@VAR=VAL[,VAR=VAL]: the library is copied to a temporary name (so the same file can load twice) and loaded -- and run once, so that
    # switches it reads once per process are read -- under those environment variables
    import os, shutil, tempfile
    path, _, envs = spec.partition("@")
    envs = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
    lib_file = path
    if envs:
        lib_file = tempfile.NamedTemporaryFile(suffix=".so", delete=False).name
        shutil.copy(path, lib_file)
    saved = {k: os.environ.get(k) for k in envs}
    os.environ.update(envs)
    path = spec
    L = C.CDLL(lib_file)
    L.savgol_create.restype = C.c_void_p
    L.savgol_create.argtypes = [C.POINTER(Cfg)]
    fn = L.savgol_apply_batch_f64 if a.f64 else L.savgol_apply_batch_f32
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_size_t] * 4 + [C.c_void_p]
    cfg = Cfg(a.n, a.m, a.deriv, 1.0, 1)
    f = L.savgol_create(C.byref(cfg))
    run = lambda fn=fn, f=f: fn(f, x.data_ptr(), y.data_ptr(), a.channels, a.length, a.length, a.length, st)
    assert run() == 0
    for k, v in saved.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    libs.append((path, run, []))
torch.cuda.synchronize()
for r in range(a.rounds):
    for path, run, ts in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
for path, run, ts in libs:
    ts = np.sort(np.array(ts))
    print(f"{path:50s} n={a.n}: median {np.median(ts):.3f} ms  min {ts[0]:.3f}  max {ts[-1]:.3f}")
