#!/usr/bin/env python3
"""A/B builds / switches of libsavgol_hip.so on the stream block push in ONE process, interleaved rounds (never rank builds across processes: the
boxes drift by 5 % within a call).   python tools/ab_stream.py lib.so lib.so@SAVGOL_HIP_STREAM_MOMENT=0 [--n 16 --m 2 --d 1 --fma 1]
Each round: K launches back to back per library (sustained: what bench.py reports), libraries in turn."""
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
ap.add_argument("--m", type=int, default=2)
ap.add_argument("--d", type=int, default=1)
ap.add_argument("--fma", type=int, default=1)
ap.add_argument("--streams", type=int, default=65536)
ap.add_argument("--ticks", type=int, default=4096)
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--burst", type=int, default=7)
a = ap.parse_args()


class Cfg(C.Structure):
    _fields_ = [("half_window", C.c_uint8), ("poly_order", C.c_uint8), ("derivative", C.c_uint8), ("time_step", C.c_float), ("boundary", C.c_int)]


x = torch.randn((a.ticks, a.streams), device="cuda")
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
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
    cfg = Cfg(a.n, a.m, a.d, 1e-3, 0)
    bank = L.savgol_streambank_create_ex(C.byref(cfg), a.streams, 1 if a.fma else 0)
    assert bank
    run = lambda L=L, bank=bank: L.savgol_streambank_push_block(bank, x.data_ptr(), a.ticks, y.data_ptr(), st)
    assert run() >= 0
    torch.cuda.synchronize()
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    libs.append((spec, run, []))
for r in range(a.rounds):
    for spec, run, ts in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.burst):
            run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / a.burst)
alg = 8.0 * a.streams * a.ticks
for spec, run, ts in libs:
    ts = np.sort(np.array(ts))
    med = float(np.median(ts))
    print(f"{spec[-70:]:70s} n={a.n} m={a.m} d={a.d} fma={a.fma}: median {med:.4f} ms = {alg / (med * 1e-3) / 8e12:.3f}  min {ts[0]:.4f}  max {ts[-1]:.4f}")
