"""A/B two builds of libsavgol_hip.so on the 2-D batch path in one process (config 4 shape).  python tools/ab_2d.py libA.so libB.so [--n 7]
A library given as path@VAR=VAL[,VAR=VAL] is copied to a temporary name (so the same file can load twice) and has those environment
variables set during its first call -- the library's knobs are read once, on first use."""
import argparse, ctypes as C, os, shutil, sys, tempfile
import numpy as np, torch
ap = argparse.ArgumentParser(); ap.add_argument("libs", nargs="+"); ap.add_argument("--n", type=int, default=7); ap.add_argument("--images", type=int, default=64); ap.add_argument("--cols", type=int, default=4096); ap.add_argument("--rows", type=int, default=4096); ap.add_argument("--boundary", type=int, default=1); ap.add_argument("--zeros", action="store_true", help="all-zero frames (data-dependent power)"); ap.add_argument("--order", type=int, default=3); ap.add_argument("--dx", type=int, default=0); ap.add_argument("--dy", type=int, default=0); ap.add_argument("--ny", type=int, default=0, help="half_window_y (default: --n, a square window)"); ap.add_argument("--method", type=int, default=2)
a = ap.parse_args()
a.ny = a.ny or a.n
class Cfg2(C.Structure):
    _fields_ = [("nx", C.c_uint8), ("ny", C.c_uint8), ("order", C.c_uint8), ("dx", C.c_uint8), ("dy", C.c_uint8), ("ddx", C.c_float), ("ddy", C.c_float)]
cols, rows = a.cols, a.rows
x = torch.zeros((a.images, rows, cols), device="cuda") if a.zeros else torch.randn((a.images, rows, cols), device="cuda"); y = torch.empty_like(x)
runs = []
for spec in a.libs:
    path, _, envs = spec.partition("@")
    envs = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
    if envs:
        tmp = tempfile.NamedTemporaryFile(suffix=".so", delete=False).name
        shutil.copy(path, tmp)
        lib_file, path = tmp, spec
    else:
        lib_file = path
    saved = {k: os.environ.get(k) for k in envs}
    os.environ.update(envs)
    L = C.CDLL(lib_file)
    L.savgol2d_create.restype = C.c_void_p; L.savgol2d_create.argtypes = [C.POINTER(Cfg2)]
    L.savgol2d_apply_batch_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    f = L.savgol2d_create(C.byref(Cfg2(a.n, a.ny, a.order, a.dx, a.dy, 1.0, 1.0)))
    run = lambda L=L, f=f: L.savgol2d_apply_batch_f32(f, x.data_ptr(), rows, cols, cols, rows * cols, y.data_ptr(), cols, rows * cols, a.images, a.boundary, a.method, None)
    assert run() == 0
    for k, v in saved.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    runs.append((path, run, []))
torch.cuda.synchronize()
for r in range(10):
    for path, run, ts in runs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
for path, run, ts in runs:
    print(f"{path:50s} n={a.n}x{a.ny} m{a.method} order={a.order} d=({a.dx},{a.dy}) {rows}x{cols}x{a.images}: median {np.median(ts):.3f} ms  min {min(ts):.3f}")
