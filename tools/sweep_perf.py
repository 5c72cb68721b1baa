"""Throughput of every kernel family across half windows: looks for performance cliffs (tools, not product)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sg = load_package()
import torch


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


which = sys.argv[1:] or ["1d", "2d"]
if "1d" in which:
    for dtype, ch in ((torch.float32, 2048), (torch.float64, 1024)):
        x = torch.randn((ch, 1 << 20), dtype=dtype, device="cuda")
        for n in (1, 2, 3, 4, 6, 8, 12, 16, 20, 23, 24, 28, 32):
            line = f"1-D {str(dtype)[6:]:8s} n={n:2d}:"
            for mode in (0, 1):
                f = sg.Filter(n, min(4, 2 * n), 0, 1.0, mode)
                ms = timed(lambda: f.apply_tensor(x))
                line += f"  mode {mode}: {ms:7.3f} ms {2 * x.numel() * x.element_size() / ms / 1e6:6.0f} GB/s"
            print(line, flush=True)
if "2d" in which:
    images, size = 16, 4096
    x = torch.randn((images, size, size), device="cuda")
    y = torch.empty_like(x)
    for n in range(1, 17):
        line = f"2-D n={n:2d}:"
        for method in (2, 1):
            f = sg.Filter2D(n, n, min(3, 2 * n))
            ms = timed(lambda: f.apply_batch(x, y, size, size, images, boundary=1, method=method), reps=2)
            line += f"  method {method}: {ms:8.3f} ms {images * size * size / ms / 1e6:7.1f} Gpix/s"
        print(line, flush=True)
    for (nx, ny) in ((4, 7), (7, 4), (2, 12)):
        f = sg.Filter2D(nx, ny, 3)
        line = f"2-D {2 * nx + 1}x{2 * ny + 1} window:"
        for method in (2, 1):
            ms = timed(lambda: f.apply_batch(x, y, size, size, images, boundary=1, method=method), reps=2)
            line += f"  method {method}: {ms:8.3f} ms {images * size * size / ms / 1e6:7.1f} Gpix/s"
        print(line, flush=True)
if "2d-orders" in which:
    # higher polynomial orders = more separable terms (order 4-5: three, order 6: four); which kernel takes them is printed by
    # SAVGOL_HIP_TRACE-less means: compare with the order-3 line of the same half window
    images, size = 16, 4096
    x = torch.randn((images, size, size), device="cuda")
    y = torch.empty_like(x)
    for n in (4, 6, 8, 9, 10, 12, 13, 14, 16):
        line = f"2-D n={n:2d}:"
        for order in (3, 4, 6):
            if order > 2 * n:
                continue
            f = sg.Filter2D(n, n, order)
            ms = timed(lambda: f.apply_batch(x, y, size, size, images, boundary=1, method=2), reps=2)
            line += f"  order {order}: {ms:8.3f} ms {images * size * size / ms / 1e6:7.1f} Gpix/s"
        print(line, flush=True)
    L = sg.lib()
    o2, o3 = torch.empty_like(x), torch.empty_like(x)
    pitch = size * size
    for n in (4, 8, 9, 12, 16):
        for order in (3, 4, 6):
            if order > 2 * n:
                continue
            g = timed(lambda: L.savgol2d_gradient_batch_f32(n, n, order, x.data_ptr(), size, size, size, pitch, y.data_ptr(), o2.data_ptr(), size, pitch, images, 1.0, 1.0, 1, None), reps=2)
            h = timed(lambda: L.savgol2d_hessian_batch_f32(n, n, order, x.data_ptr(), size, size, size, pitch, y.data_ptr(), o2.data_ptr(), o3.data_ptr(), size, pitch, images, 1.0, 1.0, 1, None), reps=2)
            l = timed(lambda: L.savgol2d_laplacian_batch_f32(n, n, order, x.data_ptr(), size, size, size, pitch, y.data_ptr(), size, pitch, images, 1.0, 1.0, 1, None), reps=2)
            singles = []
            for (dx, dy) in ((1, 0), (0, 1), (2, 0), (1, 1), (0, 2)):
                f = sg.Filter2D(n, n, order, dx, dy)
                singles.append(timed(lambda: f.apply_batch(x, y, size, size, images, boundary=1, method=2), reps=2))
            print(f"2-D n={n:2d} order {order}: gradient {g:7.3f} ms  hessian {h:7.3f}  laplacian {l:7.3f}   singles (1,0) (0,1) (2,0) (1,1) (0,2): " +
                  " ".join(f"{s:6.3f}" for s in singles), flush=True)
