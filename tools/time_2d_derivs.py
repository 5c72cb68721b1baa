"""Times the fused derivative entry points against one single-output separable launch per frame (tools, not product)."""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
sg = load_package()
import torch

images, size, n, order = 64, 4096, 7, 3
x = torch.randn((images, size, size), device="cuda")
outs = [torch.empty_like(x) for _ in range(3)]
L = sg.lib()
pitch = size * size

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for b in (0, 1):
    t = timed(lambda: L.savgol2d_gradient_batch_f32(n, n, order, x.data_ptr(), size, size, size, pitch, outs[0].data_ptr(), outs[1].data_ptr(), size, pitch, images, 1.0, 1.0, b, None))
    print(f"boundary {b}: fused gradient  {t:.3f} ms")
    t = timed(lambda: L.savgol2d_hessian_batch_f32(n, n, order, x.data_ptr(), size, size, size, pitch, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), size, pitch, images, 1.0, 1.0, b, None))
    print(f"boundary {b}: fused hessian   {t:.3f} ms")
    t = timed(lambda: L.savgol2d_laplacian_batch_f32(n, n, order, x.data_ptr(), size, size, size, pitch, outs[0].data_ptr(), size, pitch, images, 1.0, 1.0, b, None))
    print(f"boundary {b}: laplacian       {t:.3f} ms")
    for (dx, dy) in ((0, 0), (1, 0), (0, 1), (2, 0), (1, 1), (0, 2)):
        f = sg.Filter2D(n, n, order, dx, dy)
        for m in (2, 3):
            t = timed(lambda: f.apply_batch(x, outs[0], size, size, images, boundary=b, method=m))
            print(f"boundary {b}: single d=({dx},{dy}) method {m}: {t:.3f} ms")
