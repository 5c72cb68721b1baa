#!/usr/bin/env python3
"""The 2-D frame-stack launch (savgol2d_apply_batch_f32, method 2) over FRESH ALLOCATIONS inside one process, several switches of the library side
by side -- tools/placement_1d.py's question for config 4.
    python tools/placement_2d.py lib.so lib.so@SAVGOL_HIP_ROLL_TILE=0 tools/ab/lib_variant.so ... [--images 64 --size 4096 --n 7 --boundary 1 --allocations 10]"""
import argparse
import ctypes as C
import os
import shutil
import tempfile

import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--n", type=int, default=7)
ap.add_argument("--order", type=int, default=3)
ap.add_argument("--images", type=int, default=64)
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--boundary", type=int, default=1, help="0 VALID, 1 CONSTANT, 2 REFLECT")
ap.add_argument("--allocations", type=int, default=10)
ap.add_argument("--reps", type=int, default=4)
a = ap.parse_args()


class Cfg(C.Structure):
    _fields_ = [("half_window_x", C.c_uint8), ("half_window_y", C.c_uint8), ("poly_order", C.c_uint8), ("deriv_x", C.c_uint8), ("deriv_y", C.c_uint8),
                ("delta_x", C.c_float), ("delta_y", C.c_float)]


st = torch.cuda.current_stream().cuda_stream
R = a.size
x0 = torch.randn((2, R, R), device="cuda")
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
    L.savgol2d_create.restype = C.c_void_p
    L.savgol2d_create.argtypes = [C.POINTER(Cfg)]
    fn = L.savgol2d_apply_batch_f32
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    cfg = Cfg(a.n, a.n, a.order, 0, 0, 1.0, 1.0)
    f = L.savgol2d_create(C.byref(cfg))
    assert f and fn(f, x0.data_ptr(), R, R, R, R * R, y0.data_ptr(), R, R * R, 2, a.boundary, 2, st) == 0    # the switches are read here
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
    x = torch.randn((a.images, R, R), device="cuda")
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
            run = lambda: fn(f, x.data_ptr(), R, R, R, R * R, y.data_ptr(), R, R * R, a.images, a.boundary, 2, st)
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
alg = 8.0 * a.images * R * R
print("frac@median " + "  ".join(f"{alg / (q * 1e-3) / 8e12:22.4f}" for q in np.median(r, axis=0)))
