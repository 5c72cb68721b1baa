"""BASELINE.json's configurations as parity cases at (or near) their full sizes.

config 1: 1 ch x 1e6 fp32, n=5, m=3, d=0, POLYNOMIAL (the reference's own CPU-runnable case) -- host API, whole signal
config 2: covered by test_gpu_1d.py::test_full_size_config2_properties
config 3: covered by test_gpu_stream.py::test_stream_bank_config3_shape
config 4: 4096 x 4096 fp32 frames, n=7, order 3 -- crops of full-size frames vs the oracle, all modes, both kernels
config 5: fp64, n=32, d=2, 2^22-sample channels -- sampled channels vs the fp64 oracle + linearity at scale
Size-independent properties (linearity, constant/ramp preservation) cover what the oracle cannot recompute in seconds."""
import numpy as np
import pytest

from tests._util import normwise

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu(sg):
    import torch
    assert torch.cuda.is_available() and sg.device_count() > 0, sg.last_error()
    return torch


def test_config1_reference_cpu_case_through_host_api(sg, sgo, torch_gpu):
    x = sgo.synth_f32(0, 1, 1_000_000)[0]
    y = sg.Filter(5, 3, 0, 1.0, 0).apply(x)
    ref32 = sgo.Filter(5, 3).apply(x)                          # the reference's own fp32 result (bit-exact restatement)
    hi = sgo.Filter(5, 3).apply_f64(x.astype(np.float64))
    assert normwise(y, hi) < 1e-6 and np.array_equal(y.view(np.uint32), ref32.view(np.uint32))      # host-pointer call: the reference's bits
    v = sg.Filter(5, 3).apply_valid(x)
    assert v.size == 1_000_000 - 10 and normwise(v, hi[5:-5]) < 1e-6


def test_config4_full_size_frames(sg, sgo, torch_gpu):
    torch = torch_gpu
    size, images, n = 4096, 4, 7
    x = torch.empty((images * size, size), dtype=torch.float32, device="cuda")
    sg.synth(x)
    x = x.view(images, size, size)
    f = sg.Filter2D(n, n, 3)
    o = sgo.Filter2D(n, n, 3)
    rng = np.random.default_rng(3)
    crops = [(0, 0), (0, size - 200), (size - 200, 0), (size - 200, size - 200)] + [tuple(rng.integers(100, size - 300, 2)) for _ in range(3)]
    for b in range(3):
        outs = {}
        for method in (1, 2):
            out = torch.full_like(x, -2.0)
            f.apply_batch(x, out, size, size, images, boundary=b, method=method)
            outs[method] = out
        torch.cuda.synchronize()
        for k in (0, images - 1):
            xh = x[k].cpu().numpy()
            for (r0, c0) in crops:
                r0, c0 = int(r0), int(c0)
                # oracle on the crop + its halo: interior of the crop does not depend on how the crop's own edge is padded,
                # except where the crop touches the frame edge -- there the padded mode itself is what is compared
                ra, rb = max(r0 - n, 0), min(r0 + 200 + n, size)
                ca, cb = max(c0 - n, 0), min(c0 + 200 + n, size)
                sub = np.ascontiguousarray(xh[ra:rb, ca:cb])
                mode = b if b != 0 else 1
                want32 = o.apply(sub, sub.shape[1], mode)
                want64 = o.apply_f64acc(sub, sub.shape[1], mode)
                # rows/cols of the sub-result that are exact for the full frame: everything not within n of an ARTIFICIAL edge
                t = 0 if ra == 0 else n; l = 0 if ca == 0 else n
                bt = sub.shape[0] - (0 if rb == size else n); rt = sub.shape[1] - (0 if cb == size else n)
                if b == 0:                       # VALID: frame border rows are not produced at all
                    t = max(t, n - ra) if ra < n else t; l = max(l, n - ca) if ca < n else l
                    bt = min(bt, size - n - ra); rt = min(rt, size - n - ca)
                g1 = outs[1][k, ra + t:ra + bt, ca + l:ca + rt].cpu().numpy()
                g2 = outs[2][k, ra + t:ra + bt, ca + l:ca + rt].cpu().numpy()
                assert np.array_equal(g1, want32[t:bt, l:rt]), (b, k, r0, c0)                    # dense kernel: bit-exact
                assert normwise(g2, want64[t:bt, l:rt]) < 1e-6, (b, k, r0, c0)                   # separable: fp32 rounding
        if b == 0:
            assert torch.all(outs[1][:, :n] == -2.0) and torch.all(outs[2][:, :, -n:] == -2.0)   # VALID leaves the border alone
    # a constant frame stays constant (weights sum to 1) under both padded modes
    c = torch.full((1, size, size), 3.25, dtype=torch.float32, device="cuda")
    out = torch.empty_like(c)
    for b in (1, 2):
        f.apply_batch(c, out, size, size, 1, boundary=b, method=2)
        assert (out - 3.25).abs().max().item() < 1e-5


def test_config5_fp64_second_derivative_long_channels(sg, sgo, torch_gpu):
    torch = torch_gpu
    ch, length = 192, 1 << 22                                   # per-GPU slice of config 5 is 4096 such channels
    free, _ = torch.cuda.mem_get_info()
    if free < 3 * ch * length * 8 + (2 << 30):
        pytest.skip("not enough HBM free")
    x = torch.empty((ch, length), dtype=torch.float64, device="cuda")
    sg.synth(x, channel0=1000)
    f = sg.Filter(32, 4, 2, 1.0, 0)
    y = f.apply_tensor(x)
    sample = [0, 1, 95, 191]
    ref = sgo.Filter(32, 4, 2, 1.0, 0).apply_f64(x[sample].cpu().numpy())
    assert normwise(y[sample].cpu().numpy(), ref) < 1e-12
    # linearity at scale: F(a x + b t) = a F(x) + b F(t), t a ramp (the fp32 tables do not annihilate a ramp exactly,
    # so F(t) is taken from the kernel too)
    t = torch.arange(length, dtype=torch.float64, device="cuda").expand(ch, length).contiguous()
    ft = f.apply_tensor(t)
    z = f.apply_tensor(2.5 * x + 1e-3 * t)
    d = (z - (2.5 * y + 1e-3 * ft)).abs().max().item()
    assert d < 1e-10 * max(1.0, z.abs().max().item()), d


def _need_hbm(torch, nbytes):
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < nbytes + (4 << 30):
        pytest.skip(f"needs {nbytes / 2**30:.0f} GiB of free HBM, {free / 2**30:.0f} there")


def test_config4_full_batch_512_frames(sg, sgo, torch_gpu):
    """BASELINE config 4 at its full size inside pytest (VERDICT r03 missing #3): 512 frames of 4096 x 4096 fp32 (34.4 GB in + 34.4 GB
    out), n = 7, order 3, the default (fast) method, all three boundary modes.  Crops of frames 0 / 255 / 511 -- the four corners, where
    the padded mode itself is what is compared, and two interior windows -- against the double-accumulation oracle; VALID leaves the
    border of EVERY frame alone; a batch-wide checksum says no frame was skipped or written twice: the mean of a smoothed frame is the
    mean of its input to 1e-4 (weights sum to 1), per frame."""
    torch = torch_gpu
    size, images, n = 4096, 512, 7
    _need_hbm(torch, 2 * images * size * size * 4)
    x = torch.empty((images, size, size), dtype=torch.float32, device="cuda")
    for i0 in range(0, images, 64):                               # 64-frame pieces: the generator indexes 2^32 samples per call
        sg.synth(x[i0:i0 + 64].view(64 * size, size), channel0=i0 * size)
    out = torch.empty_like(x)
    f = sg.Filter2D(n, n, 3)
    o = sgo.Filter2D(n, n, 3)
    in_means = x.view(images, -1).double().mean(dim=1)
    crops = [(0, 0), (0, size - 160), (size - 160, 0), (size - 160, size - 160), (1000, 2040), (2049, 3333)]
    for b in range(3):
        out.fill_(-2.0)
        f.apply_batch(x, out, size, size, images, boundary=b, method=0)
        torch.cuda.synchronize()
        for k in (0, 255, images - 1):
            xh = x[k].cpu().numpy()
            for (r0, c0) in crops:
                ra, rb = max(r0 - n, 0), min(r0 + 160 + n, size)
                ca, cb = max(c0 - n, 0), min(c0 + 160 + n, size)
                sub = np.ascontiguousarray(xh[ra:rb, ca:cb])
                want64 = o.apply_f64acc(sub, sub.shape[1], b if b != 0 else 1)
                t = 0 if ra == 0 else n; l = 0 if ca == 0 else n
                bt = sub.shape[0] - (0 if rb == size else n); rt = sub.shape[1] - (0 if cb == size else n)
                if b == 0:
                    t = max(t, n - ra) if ra < n else t; l = max(l, n - ca) if ca < n else l
                    bt = min(bt, size - n - ra); rt = min(rt, size - n - ca)
                got = out[k, ra + t:ra + bt, ca + l:ca + rt].cpu().numpy()
                assert normwise(got, want64[t:bt, l:rt]) < 1e-6, (b, k, r0, c0)
        if b == 0:
            assert torch.all(out[:, :n] == -2.0) and torch.all(out[:, -n:] == -2.0)
            assert torch.all(out[:, :, :n] == -2.0) and torch.all(out[:, :, -n:] == -2.0)
            inner = out[:, n:-n, n:-n]
            assert not torch.any(inner == -2.0)
        else:
            assert not torch.any(out == -2.0)
            out_means = out.view(images, -1).double().mean(dim=1)
            assert torch.all((out_means - in_means).abs() < 1e-4), (out_means - in_means).abs().max().item()
    del out
    # a constant batch stays constant under both padded modes (the additive kernel's box sums must not drift down a 4096-row frame)
    x[:8].fill_(3.25)
    y = torch.empty_like(x[:8])
    for b in (1, 2):
        f.apply_batch(x[:8], y, size, size, 8, boundary=b, method=0)
        assert (y - 3.25).abs().max().item() < 1e-5


def test_config5_full_slice_in_chunks(sg, sgo, torch_gpu):
    """One GPU's slice of BASELINE config 5 at full size inside pytest (VERDICT r03 missing #3): 4096 channels x 2^22 fp64 samples,
    n = 32, m = 4, d = 2, POLYNOMIAL, processed in 1024-channel chunks exactly as bench.py's config-5 line does (input chunk resident,
    one chunk-sized output).  Oracle on channels of the first and the last chunk (1e-12), linearity on the last chunk."""
    torch = torch_gpu
    channels, chunk, length = 4096, 1024, 1 << 22
    _need_hbm(torch, 4 * chunk * length * 8)
    f = sg.Filter(32, 4, 2, 1.0, 0)
    of = sgo.Filter(32, 4, 2, 1.0, 0)
    x = torch.empty((chunk, length), dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    checked = 0
    for c0 in range(0, channels, chunk):
        sg.synth(x, channel0=c0)
        y.fill_(float("nan"))
        f.apply_batch(x, y, chunk, length, dtype="f64")
        torch.cuda.synchronize()
        assert not torch.isnan(y).any()                           # every output of every channel written (edges included)
        if c0 in (0, channels - chunk):
            sample = [0, 1, chunk // 2, chunk - 1]
            ref = of.apply_f64(x[sample].cpu().numpy())
            assert normwise(y[sample].cpu().numpy(), ref) < 1e-12, c0
            # round 6: the same chunk through savgol_apply_batch_f64_tol with the bar the config states -- what bench.py's
            # config-5 figure times: another kernel (different bits), inside 1e-6; a tolerance below that keeps the 1e-12 path (same bits)
            z = torch.full_like(y, float("nan"))
            f.apply_batch(x, z, chunk, length, dtype="f64", rel_tol=1e-6)
            torch.cuda.synchronize()
            assert not torch.isnan(z).any() and not torch.equal(z, y)
            e = normwise(z[sample].cpu().numpy(), ref)
            assert 1e-13 < e < 1e-6, (c0, e)
            f.apply_batch(x, z, chunk, length, dtype="f64", rel_tol=1e-9)
            torch.cuda.synchronize()
            assert torch.equal(z, y)
            del z
            checked += 1
    assert checked == 2
    with pytest.raises(RuntimeError, match="rel_tol"):
        f.apply_batch(x, y, chunk, length, dtype="f64", rel_tol=float("nan"))
    t = torch.arange(length, dtype=torch.float64, device="cuda")
    ft = f.apply_tensor(t.view(1, -1).contiguous())
    z = torch.empty_like(y)
    x.mul_(2.5).add_(1e-3 * t)
    f.apply_batch(x, z, chunk, length, dtype="f64")
    d = (z - (2.5 * y + 1e-3 * ft)).abs().max().item()
    assert d < 1e-10 * max(1.0, z.abs().max().item()), d


def test_batch_call_is_graph_capturable(sg, sgo, torch_gpu):
    """savgol_hip.h promises that the batch calls only enqueue (after a warm-up that uploads the tables)."""
    torch = torch_gpu
    x = torch.empty((64, 20000), dtype=torch.float32, device="cuda")
    sg.synth(x)
    y = torch.zeros_like(x)
    f = sg.Filter(32, 4, 0, 1.0, 0)
    f.apply_batch(x, y, 64, 20000)                             # warm-up: uploads the edge table
    torch.cuda.synchronize()
    want = y.clone()
    y.zero_()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            f.apply_batch(x, y, 64, 20000, stream=s)
    y.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, want)


def test_round3_entry_points_are_graph_capturable(sg, torch_gpu):
    """The entry points added in round 3 that promise "launches only": per-call flags (_ex), the fused strided call, the FMA stream
    bank's block push, and the three-output Hessian / rectangular-window launches.  Captured once, replayed, compared with the
    eager results."""
    torch = torch_gpu
    L = sg.lib()
    x = torch.empty((32, 9000), dtype=torch.float32, device="cuda")
    sg.synth(x)
    y_ex, y_ref = torch.zeros_like(x), torch.zeros_like(x)
    aos = torch.randn((4, 5000, 4), device="cuda")
    aos_out = torch.zeros_like(aos)
    f = sg.Filter(32, 4, 1, 1.0, 0)
    T, S = 200, 4096
    xs = torch.randn((T, S), device="cuda")
    ys = torch.zeros_like(xs)
    bank = sg.StreamBank(S, 16, 2, 1, 1e-3, fma=True)
    img = torch.randn((2, 160, 520), device="cuda")
    rect_out, hess = torch.zeros_like(img), [torch.zeros_like(img) for _ in range(3)]
    f2 = sg.Filter2D(4, 7, 3)

    def enqueue(stream):
        h = stream.cuda_stream if stream is not None else None
        f.apply_batch(x, y_ex, 32, 9000, stream=stream, flags=sg.SAVGOL_BATCH_PLAIN_SUMMATION | sg.SAVGOL_BATCH_CORRECT_LEADING_EDGE)
        f.apply_batch(x, y_ref, 32, 9000, stream=stream, flags=sg.SAVGOL_BATCH_REFERENCE_SUMMATION)
        assert L.savgol_apply_strided_batch_f32(f.ptr, aos.data_ptr(), 16, 4, 5000 * 16, aos_out.data_ptr(), 16, 8, 5000 * 16, 4, 5000, h) == 0
        f2.apply_batch(img, rect_out, 160, 520, 2, boundary=2, method=2, stream=stream)
        assert L.savgol2d_hessian_batch_f32(7, 7, 3, img.data_ptr(), 160, 520, 520, 160 * 520, hess[0].data_ptr(), hess[1].data_ptr(), hess[2].data_ptr(),
                                            520, 160 * 520, 2, 1.0, 1.0, 1, h) == 0

    enqueue(None)                                              # warm-up: tables, plans
    bank.push_block(xs, T, ys)
    torch.cuda.synchronize()
    want = [t.clone() for t in (y_ex, y_ref, aos_out, rect_out, *hess)]
    want_stream = ys.clone()
    bank2 = sg.StreamBank(S, 16, 2, 1, 1e-3, fma=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            enqueue(s)
            bank2.push_block(xs, T, ys, stream=s)
    for t in (y_ex, y_ref, aos_out, rect_out, *hess, ys):
        t.zero_()
    g.replay()
    torch.cuda.synchronize()
    for t, w in zip((y_ex, y_ref, aos_out, rect_out, *hess), want):
        assert torch.equal(t, w)
    assert torch.equal(ys[32:], want_stream[32:])


def test_2d_batch_and_derivative_calls_are_graph_capturable(sg, torch_gpu):
    """Same promise for the 2-D device entry points (rolling-window kernel, fused tile kernel, dense kernel)."""
    torch = torch_gpu
    images, rows, cols = 3, 200, 520
    x = torch.randn((images, rows, cols), device="cuda")
    outs = [torch.zeros_like(x) for _ in range(9)]
    f = sg.Filter2D(7, 7, 3)
    L = sg.lib()
    pitch = rows * cols

    def enqueue(stream):
        h = stream.cuda_stream if stream is not None else None
        f.apply_batch(x, outs[0], rows, cols, images, boundary=1, method=2, stream=stream)
        f.apply_batch(x, outs[1], rows, cols, images, boundary=2, method=1, stream=stream)
        assert L.savgol2d_gradient_batch_f32(7, 7, 3, x.data_ptr(), rows, cols, cols, pitch, outs[2].data_ptr(), outs[3].data_ptr(),
                                             cols, pitch, images, 1.0, 1.0, 1, h) == 0
        assert L.savgol2d_hessian_batch_f32(7, 7, 3, x.data_ptr(), rows, cols, cols, pitch, outs[4].data_ptr(), outs[5].data_ptr(),
                                            outs[6].data_ptr(), cols, pitch, images, 1.0, 1.0, 1, h) == 0
        # rectangular windows (nx != ny): round 1 synchronised and used a shared scratch frame here
        assert L.savgol2d_laplacian_batch_f32(4, 6, 3, x.data_ptr(), rows, cols, cols, pitch, outs[7].data_ptr(), cols, pitch, images,
                                              1.0, 1.0, 1, h) == 0
        assert L.savgol2d_laplacian_batch_f32(7, 7, 3, x.data_ptr(), rows, cols, cols, pitch, outs[8].data_ptr(), cols, pitch, images,
                                              1.0, 1.0, 2, h) == 0

    enqueue(None)                                              # warm-up: uploads weight / factor tables
    torch.cuda.synchronize()
    want = [o.clone() for o in outs]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            enqueue(s)
    for o in outs:
        o.zero_()
    g.replay()
    torch.cuda.synchronize()
    for o, w in zip(outs, want):
        assert torch.equal(o, w)


def test_one_channel_longer_than_2_pow_32_samples(sg, sgo, torch_gpu):
    """Maximum sizes: a single channel of 2^32 + 4099 samples (17 GB in, 17 GB out) -- every sample index, tile index and byte
    offset past 32 bits.  The convolution is local, so windows of the output are checked against the oracle run on the
    matching slices of the input: both ends (edge rows), and around 2^31 and 2^32 samples (4 GiB / 8 GiB / 16 GiB of bytes)."""
    torch = torch_gpu
    length = (1 << 32) + 4099
    free, _ = torch.cuda.mem_get_info()
    if free < 2 * length * 4 + (4 << 30):
        pytest.skip("not enough HBM free for a 2^32-sample channel")
    x = torch.empty((1, length), dtype=torch.float32, device="cuda")
    sg.synth(x)
    y = torch.empty_like(x)
    n = 32
    for mode in (0, 2):                                         # POLYNOMIAL edge rows, PERIODIC wrap (index length-1 ... 0)
        f = sg.Filter(n, 4, 0, 1.0, mode)
        f.apply_batch(x, y, 1, length)
        torch.cuda.synchronize()
        # interior windows: oracle on [c - 2000 - n, c + 2000 + n), compared on the inner 4000 samples
        for c in ((1 << 30), (1 << 31) - 7, (1 << 31) + 2048 * 3 + 5, (1 << 32) - 1, (1 << 32) + 2000):
            lo, hi = c - 2000 - n, min(c + 2000 + n, length)
            xs = x[0, lo:hi].cpu().numpy().astype(np.float64)[None]
            ref = sgo.Filter(n, 4, 0, 1.0, 0).apply_f64(xs)[0][n:-n]
            got = y[0, lo + n:hi - n].cpu().numpy()
            assert normwise(got, ref) < 1e-6, (mode, c, normwise(got, ref))
        if mode == 0:
            # the two ends: the oracle on the first / last 4096 samples gives the same edge rows as on the whole channel
            for sl in (slice(0, 4096), slice(length - 4096, length)):
                xs = x[0, sl].cpu().numpy().astype(np.float64)[None]
                ref = sgo.Filter(n, 4, 0, 1.0, 0).apply_f64(xs)[0]
                got = y[0, sl].cpu().numpy()
                keep = slice(0, 4096 - n) if sl.start == 0 else slice(n, 4096)
                assert normwise(got[keep], ref[keep]) < 1e-6
        else:
            # PERIODIC: the first n outputs read the last n samples of the channel and vice versa
            ring = torch.cat((x[0, length - 2048:], x[0, :2048])).cpu().numpy().astype(np.float64)[None]
            ref = sgo.Filter(n, 4, 0, 1.0, 0).apply_f64(ring)[0]
            got = torch.cat((y[0, length - 2048:], y[0, :2048])).cpu().numpy()
            assert normwise(got[n:-n], ref[n:-n]) < 1e-6
    # the reference's summation order on the same channel: bit-identical to the oracle's fp32 path, REFLECT edges included
    L = sg.lib()
    assert L.savgol_hip_set_option(sg.SAVGOL_HIP_OPT_REFERENCE_SUMMATION, 1) == 0
    try:
        f = sg.Filter(n, 4, 0, 1.0, 1)
        f.apply_batch(x, y, 1, length)
        torch.cuda.synchronize()
    finally:
        L.savgol_hip_set_option(sg.SAVGOL_HIP_OPT_REFERENCE_SUMMATION, 0)
    for c in ((1 << 29) + n, (1 << 30) + n - 1, (1 << 32) + 17):
        lo, hi = c - 1000 - n, c + 1000 + n
        ref = sgo.Filter(n, 4, 0, 1.0, 0).apply(x[0, lo:hi].cpu().numpy())[n:-n]
        assert np.array_equal(y[0, lo + n:hi - n].cpu().numpy().view(np.uint32), ref.view(np.uint32)), c
    for sl, keep in ((slice(0, 4096), slice(0, 4096 - n)), (slice(length - 4096, length), slice(n, 4096))):
        ref = sgo.Filter(n, 4, 0, 1.0, 1).apply(x[0, sl].cpu().numpy())
        assert np.array_equal(y[0, sl].cpu().numpy()[keep].view(np.uint32), ref[keep].view(np.uint32))
    # VALID on the long channel: out[j] = full[j + n]
    v = torch.empty((1, length - 2 * n), dtype=torch.float32, device="cuda")
    sg.Filter(n, 4, 0, 1.0, 0).apply_batch(x, v, 1, length, valid=True)
    sg.Filter(n, 4, 0, 1.0, 0).apply_batch(x, y, 1, length)
    torch.cuda.synchronize()
    for c in (0, (1 << 29) - 5, (1 << 31), length - 2 * n - 3000):
        assert torch.equal(v[0, c:c + 3000], y[0, c + n:c + n + 3000]), c
    del x, y, v
    torch.cuda.empty_cache()


def test_fp64_channels_longer_than_one_launch(sg, sgo, torch_gpu):
    """Three fp64 channels of 2^30 + 2^29 + 77 samples with an odd row pitch (unaligned rows): the sub-row path of the fp64 kernel."""
    torch = torch_gpu
    length, ld, ch, n = (1 << 30) + (1 << 29) + 77, (1 << 30) + (1 << 29) + 79, 3, 32
    free, _ = torch.cuda.mem_get_info()
    if free < 2 * ch * ld * 8 + (4 << 30):
        pytest.skip("not enough HBM free")
    xb = torch.empty((ch, ld), dtype=torch.float64, device="cuda")
    sg.synth(xb)
    yb = torch.zeros_like(xb)
    f = sg.Filter(n, 4, 2, 1.0, 3)                                  # CONSTANT edges, second derivative
    f.apply_batch(xb, yb, ch, length, in_ld=ld, out_ld=ld, dtype="f64")
    torch.cuda.synchronize()
    o = sgo.Filter(n, 4, 2, 1.0, 3)
    for c in (1, 2):
        for mid in ((1 << 29) + n, (1 << 30) + n + 3, length - 3000):
            lo, hi = mid - 1500 - n, min(mid + 1500 + n, length)
            ref = o.apply_f64(xb[c, lo:hi].cpu().numpy()[None])[0][n:-n]
            assert normwise(yb[c, lo + n:hi - n].cpu().numpy(), ref) < 1e-12, (c, mid)
        for sl, keep in ((slice(0, 4096), slice(0, 4096 - n)), (slice(length - 4096, length), slice(n, 4096))):
            ref = o.apply_f64(xb[c, sl].cpu().numpy()[None])[0]
            assert normwise(yb[c, sl].cpu().numpy()[keep], ref[keep]) < 1e-12
        assert torch.count_nonzero(yb[c, length:]).item() == 0      # the pad between rows is not written
    del xb, yb
    torch.cuda.empty_cache()


def test_host_pointer_call_on_a_signal_longer_than_one_launch(sg, sgo, torch_gpu):
    """savgol_apply / savgol_apply_valid on 2^30 + 12345 host samples (the reference takes a size_t length): bit-identical to the
    oracle on windows across the segment boundaries and at both ends."""
    import psutil
    length, n = (1 << 30) + 12345, 5
    if psutil.virtual_memory().available < 3 * length * 4 + (8 << 30):
        pytest.skip("not enough host memory")
    x = np.empty(length, np.float32)
    blk = 1 << 24
    rng = np.random.default_rng(7)
    for a in range(0, length, blk):                                  # cheap deterministic fill, block by block
        b = min(a + blk, length)
        x[a:b] = np.sin(np.arange(a, b, dtype=np.float64) * 1e-3).astype(np.float32) + rng.random(b - a, np.float32) * 0.1
    y = np.zeros_like(x)
    f = sg.Filter(n, 3, 0, 1.0, 0)
    f.apply(x, out=y)
    o = sgo.Filter(n, 3, 0, 1.0, 0)
    for c in (3000, (1 << 29) + n, (1 << 30) - 3, length - 3000):
        lo, hi = c - 2000 - n, c + 2000 + n
        ref = o.apply(x[lo:hi])[n:-n]
        assert np.array_equal(y[lo + n:hi - n].view(np.uint32), ref.view(np.uint32)), c
    for sl, keep in ((slice(0, 4096), slice(0, 4096 - n)), (slice(length - 4096, length), slice(n, 4096))):
        ref = o.apply(x[sl])
        assert np.array_equal(y[sl][keep].view(np.uint32), ref[keep].view(np.uint32))
    v = f.apply_valid(x)
    assert v.size == length - 2 * n
    for c in (0, (1 << 29) - 7, length - 2 * n - 5000):
        assert np.array_equal(v[c:c + 5000], y[c + n:c + n + 5000])


def test_one_frame_with_more_than_2_pow_31_pixels(sg, sgo, torch_gpu):
    """Maximum sizes in 2-D: one 47000 x 46500 frame on a 46504 pitch (2.19e9 pixels, pixel offsets past 2^31, byte offsets past 2^33),
    REFLECT border, both the bit-exact dense kernel and the separable one, checked on crops at the corners and deep inside."""
    torch = torch_gpu
    rows, cols, stride, n = 47000, 46500, 46504, 5
    free, _ = torch.cuda.mem_get_info()
    if free < 3 * rows * stride * 4 + (4 << 30):
        pytest.skip("not enough HBM free")
    x = torch.empty((rows, stride), dtype=torch.float32, device="cuda")
    sg.synth(x)
    f = sg.Filter2D(n, n, 3)
    o = sgo.Filter2D(n, n, 3)
    crops = [(0, 0), (0, cols - 150), (rows - 150, 0), (rows - 150, cols - 150), (46180, 46200), (23000, 40000), (46700, 7)]
    for method in (1, 2):
        out = torch.full_like(x, -2.0)
        f.apply_batch(x, out, rows, cols, 1, boundary=2, method=method, in_stride=stride, out_stride=stride)
        torch.cuda.synchronize()
        for (r0, c0) in crops:
            ra, rb = max(r0 - n, 0), min(r0 + 150 + n, rows)
            ca, cb = max(c0 - n, 0), min(c0 + 150 + n, cols)
            sub = np.ascontiguousarray(x[ra:rb, ca:cb].cpu().numpy())
            t = 0 if ra == 0 else n; l = 0 if ca == 0 else n
            bt = sub.shape[0] - (0 if rb == rows else n); rt = sub.shape[1] - (0 if cb == cols else n)
            got = out[ra + t:ra + bt, ca + l:ca + rt].cpu().numpy()
            if method == 1:
                assert np.array_equal(got, o.apply(sub, sub.shape[1], 2)[t:bt, l:rt]), (method, r0, c0)
            else:
                assert normwise(got, o.apply_f64acc(sub, sub.shape[1], 2)[t:bt, l:rt]) < 1e-6, (method, r0, c0)
        assert torch.all(out[:, cols:] == -2.0)                     # the pitch padding is not written
        del out
    del x
    torch.cuda.empty_cache()
