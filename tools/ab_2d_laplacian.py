"""A/B two builds of libsavgol_hip.so on savgol2d_laplacian_batch_f32 (rectangular or square windows) in one process.
   python tools/ab_2d_laplacian.py libA.so libB.so --nx 7 --ny 3 [--order 3 --images 16]      prints ms per call and whether the two outputs are the same bits"""
import argparse, ctypes as C
import numpy as np, torch
ap = argparse.ArgumentParser(); ap.add_argument("libs", nargs="+"); ap.add_argument("--nx", type=int, default=7); ap.add_argument("--ny", type=int, default=3)
ap.add_argument("--order", type=int, default=3); ap.add_argument("--images", type=int, default=16); ap.add_argument("--size", type=int, default=4096); ap.add_argument("--boundary", type=int, default=1)
a = ap.parse_args()
x = torch.randn((a.images, a.size, a.size), device="cuda")
outs, runs = [], []
for path in a.libs:
    L = C.CDLL(path)
    L.savgol2d_laplacian_batch_f32.argtypes = [C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_void_p]
    y = torch.zeros_like(x)
    run = lambda L=L, y=y: L.savgol2d_laplacian_batch_f32(a.nx, a.ny, a.order, x.data_ptr(), a.size, a.size, a.size, a.size * a.size, y.data_ptr(), a.size, a.size * a.size, a.images, 0.5, 2.0, a.boundary, None)
    assert run() == 0
    outs.append(y); runs.append((path, run, []))
torch.cuda.synchronize()
for r in range(8):
    for path, run, ts in runs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
for path, run, ts in runs:
    print(f"{path:50s} laplacian {2 * a.nx + 1}x{2 * a.ny + 1} order {a.order} {a.images}x{a.size}^2: median {np.median(ts):.3f} ms  min {min(ts):.3f}")
if len(outs) == 2:
    print("same bits:", bool(torch.equal(outs[0], outs[1])))
