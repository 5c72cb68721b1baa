import sys, time; sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
sg = load_package(); import torch
L = sg.lib()
for n, order in ((3, 2), (7, 3), (16, 3), (16, 6)):
    size = 128
    x = torch.randn((1, size, size), device="cuda"); o = [torch.empty_like(x) for _ in range(3)]
    pitch = size * size
    def g(): return L.savgol2d_gradient_batch_f32(n, n, order, x.data_ptr(), size, size, size, pitch, o[0].data_ptr(), o[1].data_ptr(), size, pitch, 1, 1.0, 1.0, 1, None)
    def h(): return L.savgol2d_hessian_batch_f32(n, n, order, x.data_ptr(), size, size, size, pitch, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), size, pitch, 1, 1.0, 1.0, 1, None)
    def l(): return L.savgol2d_laplacian_batch_f32(n, n, order, x.data_ptr(), size, size, size, pitch, o[0].data_ptr(), size, pitch, 1, 1.0, 1.0, 1, None)
    f = sg.Filter2D(n, n, order)
    def a(): f.apply_batch(x, o[0], size, size, 1, boundary=1, method=2); return 0
    for name, fn in (("apply", a), ("gradient", g), ("hessian", h), ("laplacian", l)):
        assert fn() == 0; torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): fn()
        t1 = time.perf_counter(); torch.cuda.synchronize()
        print(f"n={n:2d} order={order} {name:9s}: {(t1 - t0) / 50 * 1e6:8.1f} us per call on the host (enqueue only)")
