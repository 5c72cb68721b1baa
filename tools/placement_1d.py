#!/usr/bin/env python3
"""The 1-D batch launch over FRESH ALLOCATIONS inside one process (old buffers kept alive, so new physical pages back the new ones), several
switches of the library side by side (each spec is loaded once, `path@VAR=VAL[,VAR=VAL]`): which part of the run-to-run spread is the physical
placement of the two buffers, and does a tile order escape it?
    python tools/placement_1d.py lib.so lib.so@SAVGOL_HIP_1D_XCD_CHUNK_LOG2=6 ... [--channels 2048 --allocations 10 --n 32 --m 4 --f64]"""
import argparse
import ctypes as C
import os
import shutil
import tempfile

import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--m", type=int, default=4)
ap.add_argument("--deriv", type=int, default=0)
ap.add_argument("--channels", type=int, default=2048)
ap.add_argument("--length", type=int, default=1 << 20)
ap.add_argument("--allocations", type=int, default=10)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--f64", action="store_true")
ap.add_argument("--flags", type=int, default=None, help="SAVGOL_BATCH_* flags: the *_ex entry point (64 = SAVGOL_BATCH_MOMENT_F64)")
a = ap.parse_args()


class Cfg(C.Structure):
    _fields_ = [("half_window", C.c_uint8), ("poly_order", C.c_uint8), ("derivative", C.c_uint8), ("time_step", C.c_float), ("boundary", C.c_int)]


dt = torch.float64 if a.f64 else torch.float32
st = torch.cuda.current_stream().cuda_stream
x0 = torch.randn((64, a.length), dtype=dt, device="cuda")
y0 = torch.empty_like(x0)
libs = []
for spec in a.libs:
    path, _, envs = spec.partition("@")
    envs = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
    lib_file = path
    if envs:
        lib_file = tempfile.NamedTemporaryFile(suffix=".so", delete=False).name
        shutil.copy(path, lib_file)
    saved = {k: os.environ.get(k) for k in envs}
    os.environ.update(envs)
    L = C.CDLL(lib_file)
    L.savgol_create.restype = C.c_void_p
    L.savgol_create.argtypes = [C.POINTER(Cfg)]
    if a.flags is None:
        fn = L.savgol_apply_batch_f64 if a.f64 else L.savgol_apply_batch_f32
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_size_t] * 4 + [C.c_void_p]
    else:
        ex = L.savgol_apply_batch_f64_ex if a.f64 else L.savgol_apply_batch_f32_ex
        ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_size_t] * 4 + [C.c_uint, C.c_void_p]
        fn = lambda f, x, y, ch, L_, a1, a2, st, ex=ex: ex(f, x, y, ch, L_, a1, a2, a.flags, st)
    cfg = Cfg(a.n, a.m, a.deriv, 1.0, 1)
    f = L.savgol_create(C.byref(cfg))
    assert f and fn(f, x0.data_ptr(), y0.data_ptr(), 64, a.length, a.length, a.length, st) == 0           # the switches are read here
    torch.cuda.synchronize()
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    libs.append((",".join(f"{k.replace('SAVGOL_HIP_', '')}={v}" for k, v in envs.items()) or "default", fn, f))
keep = []
print("allocation  " + "  ".join(f"{name:>22s}" for name, _, _ in libs) + "        copy")
rows, firsts = [], []
for i in range(a.allocations):
    x = torch.randn((a.channels, a.length), dtype=dt, device="cuda")
    y = torch.empty_like(x)
    keep += [x, y]
    # two passes over the columns, starting at a different column for every allocation; the SECOND pass is reported (the first kernel over a fresh
    # pair is slower whatever it is -- round 5 first read that as a property of the tile order in column 0)
    row = [0.0] * len(libs)
    first_pass = [0.0] * len(libs)
    order = [(i + k) % len(libs) for k in range(len(libs))]
    for pass_no in range(2):
        for j in order:
            name, fn, f = libs[j]
            run = lambda: fn(f, x.data_ptr(), y.data_ptr(), a.channels, a.length, a.length, a.length, st)
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                run()
            e1.record(); torch.cuda.synchronize()
            row[j] = e0.elapsed_time(e1) / a.reps
            if pass_no == 0:
                first_pass[j] = row[j]
    y.copy_(x); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    row.append(e0.elapsed_time(e1) / a.reps)
    rows.append(row)
    firsts.append(first_pass)
    print(f"{i:10d}  " + "  ".join(f"{v:22.4f}" for v in row), flush=True)
r = np.array(rows)
print('pass 1 med  ' + '  '.join(f'{v:22.4f}' for v in np.median(np.array(firsts), axis=0)) + '   (first pass over each fresh pair: not in the rows above)')
for label, v in (("median", np.median(r, axis=0)), ("min", r.min(axis=0)), ("max", r.max(axis=0))):
    print(f"{label:10s}  " + "  ".join(f"{q:22.4f}" for q in v))
alg = (16.0 if a.f64 else 8.0) * a.channels * a.length
print("frac@median " + "  ".join(f"{alg / (q * 1e-3) / 8e12:22.4f}" for q in np.median(r, axis=0)))
