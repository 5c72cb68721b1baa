#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench line): streaming (BASELINE config 3) and 2-D (config 4).

    python tools/bench_paths.py stream [--streams 65536 --ticks 4096]
    python tools/bench_paths.py image  [--images 64 --size 4096 --method 1]
Prints one JSON object per workload; HIP-event timing on the launch stream."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

sg = load_package()
PEAK = 8000.0


def cpu_reference(kind):
    """The reference's own code (oracle/_ref/libsavgol_ref.so, gcc -O2, 1 thread) on a bounded sample of the same
    workload, timed on this host; falls back to the oracle port if the compiled reference did not travel."""
    import ctypes as C
    from oracle import sgo
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libsavgol_ref.so")
    have_ref = os.path.exists(ref)
    if have_ref:
        from tests.golden import make_golden as mg
        L = mg.load()
    if kind == "stream":
        exe = os.path.join(os.path.dirname(ref), "cpu_stream_bench")
        if os.path.exists(exe):
            import subprocess
            n_push = 20_000_000
            v = float(subprocess.run([exe, "16", "2", "1", "0.001", str(n_push)], capture_output=True, text=True, check=True).stdout)
            return {"kind": "reference", "cores": 1, "unit": "Msamples/s", "value": v,
                    "sample": f"the reference's savgol_stream_push in a C loop, 1 stream, {n_push} samples, best of 5 (oracle/cpu_stream_bench.c)"}
        n_push = 2_000_000
        x = sgo.synth_f32(0, 1, n_push)[0]
        f = sgo.Filter(16, 2, 1, 1e-3)
        o = sgo.Stream(f)
        t0 = time.perf_counter()
        for v in x[:200000]:
            o.push(v)
        el = time.perf_counter() - t0
        return {"kind": "port", "cores": 1, "unit": "Msamples/s", "value": round(200000 / el / 1e6, 3),
                "sample": "oracle push loop through ctypes (call-overhead bound)"}
    size = 1024
    img = sgo.synth_f32(0, size, size)
    out = np.zeros_like(img)
    res = {}
    for name, b in (("VALID", 0), ("CONSTANT", 1), ("REFLECT", 2)):
        if have_ref:
            cfg = mg.Cfg2(7, 7, 3, 0, 0, 1.0, 1.0)
            f = L.savgol2d_create(C.byref(cfg))
            t0 = time.perf_counter()
            L.savgol2d_apply(f, mg.fptr(img), size, size, size, mg.fptr(out), size, b)
            el = time.perf_counter() - t0
        else:
            f = sgo.Filter2D(7, 7, 3)
            t0 = time.perf_counter(); f.apply(img, size, b); el = time.perf_counter() - t0
        res[name] = round(size * size / el / 1e6, 2)
    return {"kind": "reference" if have_ref else "port", "cores": 1, "unit": "Mpix/s", "value": res,
            "sample": f"savgol2d_apply on one {size}x{size} fp32 frame, n=7, order 3, per boundary mode"}


def ev():
    return torch.cuda.Event(enable_timing=True)


def bench_stream(a):
    S, T, n = a.streams, a.ticks, 16
    x = torch.empty((T, S), dtype=torch.float32, device="cuda")
    sg.synth(x)
    out = torch.empty((T, S), dtype=torch.float32, device="cuda")
    bank = sg.StreamBank(S, n, 2, 1, 1e-3)
    # (a) per-tick launches: wall latency per tick measured on the host around launch + sync
    o1 = torch.empty(S, dtype=torch.float32, device="cuda")
    for t in range(64):
        bank.push(x[t], o1)
    torch.cuda.synchronize()
    lat = []
    for t in range(64, 64 + 2000):
        t0 = time.perf_counter()
        bank.push(x[t % T], o1)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e6)
    lat = np.sort(np.array(lat))
    # (b) back-to-back ticks without host sync (device time per tick)
    e0, e1 = ev(), ev()
    e0.record()
    for t in range(1000):
        bank.push(x[t % T], o1)
    e1.record(); torch.cuda.synchronize()
    tick_us = e0.elapsed_time(e1)
    # (c) block push: T ticks in one launch, ring in LDS
    bank2 = sg.StreamBank(S, n, 2, 1, 1e-3)
    bank2.push_block(x, T, out); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = ev(), ev()
        e0.record(); bank2.push_block(x, T, out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    samples = S * T
    print(json.dumps({
        "workload": f"BASELINE config 3: {S} streams, n=16, m=2, d=1, dt=1e-3",
        "per_tick_launch": {"wall_latency_us_p50": round(float(lat[len(lat) // 2]), 2), "wall_latency_us_p99": round(float(lat[int(len(lat) * 0.99)]), 2),
                            "device_us_per_tick_back_to_back": round(tick_us, 3), "ns_per_sample": round(tick_us * 1e3 / S, 4),
                            "Msamples_per_s": round(S / tick_us, 1)},
        "block_push": {"ticks_per_launch": T, "ms": round(ms, 3), "ns_per_sample": round(ms * 1e6 / samples, 5),
                       "Msamples_per_s": round(samples / ms / 1e3, 1),
                       "roofline": {"bound": "hbm", "achieved": round(8.0 * samples / ms / 1e6, 1), "peak": PEAK, "unit": "GB/s",
                                    "frac": round(8.0 * samples / ms / 1e6 / PEAK, 4), "algorithmic_bytes_per_sample": 8}},
        "cpu_baseline": cpu_reference("stream"),
    }))


def bench_image(a):
    N, size, n = a.images, a.size, 7
    x = torch.empty((N * size, size), dtype=torch.float32, device="cuda")
    sg.synth(x)
    y = torch.empty_like(x)
    f = sg.Filter2D(n, n, 3)
    res = {}
    for name, b in (("VALID", 0), ("CONSTANT", 1), ("REFLECT", 2)):
        f.apply_batch(x, y, size, size, N, boundary=b, method=a.method); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = ev(), ev()
            e0.record(); f.apply_batch(x, y, size, size, N, boundary=b, method=a.method); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts))
        pix = N * size * size
        res[name] = {"ms": round(ms, 3), "Mpix_per_s": round(pix / ms / 1e3, 1),
                     "roofline": {"bound": "hbm", "achieved": round(8.0 * pix / ms / 1e6, 1), "peak": PEAK, "unit": "GB/s",
                                  "frac": round(8.0 * pix / ms / 1e6 / PEAK, 4), "algorithmic_bytes_per_pixel": 8}}
    print(json.dumps({"workload": f"BASELINE config 4 (subset): {N} images x {size}x{size} fp32, n=7, order 3, method {a.method}", "modes": res,
                      "cpu_baseline": cpu_reference("image")}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["stream", "image"])
    ap.add_argument("--streams", type=int, default=65536)
    ap.add_argument("--ticks", type=int, default=4096)
    ap.add_argument("--images", type=int, default=64)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--method", type=int, default=1)
    a = ap.parse_args()
    bench_stream(a) if a.what == "stream" else bench_image(a)
