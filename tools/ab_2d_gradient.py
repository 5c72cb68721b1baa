"""A/B builds of libsavgol_hip.so on savgol2d_gradient_batch_f32 (the fused two-output walk / tiles) in one process.
   python tools/ab_2d_gradient.py libA.so libB.so ... --n 9 [--order 2 --images 64]"""
import argparse, ctypes as C
import numpy as np, torch
ap = argparse.ArgumentParser(); ap.add_argument("libs", nargs="+"); ap.add_argument("--n", type=int, default=9); ap.add_argument("--order", type=int, default=2)
ap.add_argument("--images", type=int, default=64); ap.add_argument("--size", type=int, default=4096); ap.add_argument("--boundary", type=int, default=1)
a = ap.parse_args()
x = torch.randn((a.images, a.size, a.size), device="cuda")
runs, outs = [], []
import os, shutil, tempfile
for spec in a.libs:
    # path@VAR=VAL[,VAR=VAL]: a private copy of the library with those environment variables set during its first call (its knobs are read once)
    path, _, envs = spec.partition("@")
    envs = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
    lib_file = path
    if envs:
        lib_file = tempfile.NamedTemporaryFile(suffix=".so", delete=False).name
        shutil.copy(path, lib_file)
    os.environ.update(envs)
    L = C.CDLL(lib_file)
    path = spec
    L.savgol2d_gradient_batch_f32.argtypes = [C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_void_p]
    gx, gy = torch.zeros_like(x), torch.zeros_like(x)
    run = lambda L=L, gx=gx, gy=gy: L.savgol2d_gradient_batch_f32(a.n, a.n, a.order, x.data_ptr(), a.size, a.size, a.size, a.size * a.size, gx.data_ptr(), gy.data_ptr(), a.size, a.size * a.size, a.images, 1.0, 1.0, a.boundary, None)
    assert run() == 0
    for k in envs:
        os.environ.pop(k, None)
    runs.append((path, run, [])); outs.append((gx, gy))
torch.cuda.synchronize()
for r in range(8):
    for path, run, ts in runs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
for path, run, ts in runs:
    print(f"{path:50s} gradient n={a.n} order {a.order} {a.images}x{a.size}^2: median {np.median(ts):.3f} ms  min {min(ts):.3f}")
if len(outs) >= 2:
    print("same bits as the first:", [bool(torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])) for o in outs[1:]])
