#!/usr/bin/env python3
"""The stream block push over FRESH ALLOCATIONS inside one process (old buffers kept alive, so new physical pages back the new ones), several
switches of the library side by side: which part of the box-to-box spread is the physical placement of the two buffers, and does any tile order
escape it?   python tools/placement_stream.py lib.so lib.so@SAVGOL_HIP_STREAM_MOMENT=0 tools/ab/lib_variant.so ... [--allocations 10 --fma 1]"""
import argparse
import ctypes as C
import os
import shutil
import tempfile

import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--n", type=int, default=16)
ap.add_argument("--fma", type=int, default=1)
ap.add_argument("--streams", type=int, default=65536)
ap.add_argument("--ticks", type=int, default=4096)
ap.add_argument("--allocations", type=int, default=10)
ap.add_argument("--burst", type=int, default=7)
a = ap.parse_args()


class Cfg(C.Structure):
    _fields_ = [("half_window", C.c_uint8), ("poly_order", C.c_uint8), ("derivative", C.c_uint8), ("time_step", C.c_float), ("boundary", C.c_int)]


st = torch.cuda.current_stream().cuda_stream
x0 = torch.randn((a.ticks, a.streams), device="cuda")
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
    L.savgol_streambank_create_ex.restype = C.c_void_p
    L.savgol_streambank_create_ex.argtypes = [C.POINTER(Cfg), C.c_size_t, C.c_uint]
    L.savgol_streambank_push_block.restype = C.c_int
    L.savgol_streambank_push_block.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    cfg = Cfg(a.n, 2, 1, 1e-3, 0)
    bank = L.savgol_streambank_create_ex(C.byref(cfg), a.streams, 1 if a.fma else 0)
    assert bank and L.savgol_streambank_push_block(bank, x0.data_ptr(), a.ticks, y0.data_ptr(), st) >= 0     # the switches are read here
    torch.cuda.synchronize()
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    libs.append((envs and ",".join(f"{k.replace('SAVGOL_HIP_STREAM_', '')}={v}" for k, v in envs.items()) or "default", L, bank))
keep = [x0, y0]
print("allocation  " + "  ".join(f"{name:>14s}" for name, _, _ in libs) + "    copy")
rows, firsts = [], []
for i in range(a.allocations):
    x = torch.randn((a.ticks, a.streams), device="cuda")
    y = torch.empty_like(x)
    keep += [x, y]
    # two passes over the columns, starting at a different column for every allocation; the SECOND pass is reported
    row = [0.0] * len(libs)
    first_pass = [0.0] * len(libs)
    order = [(i + k) % len(libs) for k in range(len(libs))]
    for pass_no in range(2):
        for j in order:
            name, L, bank = libs[j]
            L.savgol_streambank_push_block(bank, x.data_ptr(), a.ticks, y.data_ptr(), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.burst):
                L.savgol_streambank_push_block(bank, x.data_ptr(), a.ticks, y.data_ptr(), st)
            e1.record(); torch.cuda.synchronize()
            row[j] = e0.elapsed_time(e1) / a.burst
            if pass_no == 0:
                first_pass[j] = row[j]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    y.copy_(x); torch.cuda.synchronize()
    e0.record()
    for _ in range(a.burst):
        y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    row.append(e0.elapsed_time(e1) / a.burst)
    rows.append(row)
    firsts.append(first_pass)
    print(f"{i:10d}  " + "  ".join(f"{v:14.4f}" for v in row))
r = np.array(rows)
print('pass 1 med  ' + '  '.join(f'{v:14.4f}' for v in np.median(np.array(firsts), axis=0)) + '   (first pass over each fresh pair: not in the rows above)')
print("median      " + "  ".join(f"{v:14.4f}" for v in np.median(r, axis=0)))
print("min         " + "  ".join(f"{v:14.4f}" for v in r.min(axis=0)))
print("max         " + "  ".join(f"{v:14.4f}" for v in r.max(axis=0)))
print("best-of-row " + f"{np.median(r[:, :-1].min(axis=1)):.4f} (median over allocations of the fastest switch per allocation)")
