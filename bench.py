#!/usr/bin/env python3
"""bench.py -- headline benchmark: Msamples/s filtered on the 1-D batch path (BASELINE.json config 2:
4096 channels x 2^20 fp32 samples per GPU, half_window=32, poly_order=4, all four boundary modes),
with the HBM-roofline fraction of the dominant kernel and the reference's CPU path timed beside it.

    python bench.py [--gpus N --steps K --warmup W]
    python bench.py --workload stream | image        (secondary: BASELINE configs 3 / 4 on one GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of savgol_apply_batch_f32 over the whole resident batch in EACH of the four boundary
modes (4 launches of the centre kernel + the tiny polynomial edge kernel).  Inputs are generated in HBM
before the timed region.  Channels are independent: with N GPUs every rank owns its own 4096 channels
(weak scaling, no data-path collective); rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
N, M, D = 32, 4, 0


def cpu_baseline(length, deriv=D, budget_s=12.0):
    """The reference's own savgol_apply (oracle/_ref, gcc -O2, 1 thread -- the reference has no threading)
    on a bounded sample of the same workload: as many 2^20-sample channels of config 2 as fit in ~12 s."""
    import ctypes as C
    from oracle import sgo
    x = sgo.synth_f32(0, 1, length)[0]
    y = np.empty_like(x)
    ref_lib = os.path.join(ROOT, "oracle", "_ref", "libsavgol_ref.so")
    if os.path.exists(ref_lib):
        from tests.golden import make_golden as mg
        L = mg.load()
        cfg = mg.Cfg(N, M, deriv, 1.0, 0)
        f = L.savgol_create(C.byref(cfg))
        run = lambda: L.savgol_apply(f, mg.fptr(x), mg.fptr(y), length)
        kind = "reference"
    else:
        f = sgo.Filter(N, M, deriv)
        run = lambda: f.apply(x)
        kind = "port"
    run()
    n_done, t0 = 0, time.perf_counter()
    while True:
        run(); n_done += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    out = {"value": round(n_done * length / el / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": kind,
           "sample": f"{n_done} channels x {length} fp32 samples (the reference is fp32 only; n={N}, m={M}, d={deriv}, POLYNOMIAL), "
                     f"savgol_apply back to back for {el:.1f} s, 1 thread"}
    # the same call from one thread per host core, each on its own channel (the reference has no threading
    # of its own; savgol_apply is re-entrant on a shared const filter and ctypes drops the GIL around it)
    if kind == "reference":
        import threading
        cores = os.cpu_count() or 1
        bufs = [(x.copy(), np.empty_like(x)) for _ in range(cores)]
        counts, stop = [0] * cores, time.perf_counter() + budget_s / 2

        def worker(i):
            xi, yi = bufs[i]
            while time.perf_counter() < stop:
                L.savgol_apply(f, mg.fptr(xi), mg.fptr(yi), length)
                counts[i] += 1
        t0 = time.perf_counter()
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(cores)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        el2 = time.perf_counter() - t0
        out["all_cores"] = {"cores": cores, "value": round(sum(counts) * length / el2 / 1e6, 2), "unit": "Msamples/s",
                            "sample": f"{sum(counts)} channels in {el2:.1f} s, one thread per host core"}
    return out


# ---------------------------------------------------------------------------------------------------------------
# secondary workloads (python bench.py --workload stream | image): BASELINE configs 3 and 4, single GPU
# ---------------------------------------------------------------------------------------------------------------
def cpu_reference(kind):
    """The reference's own code (oracle/_ref/libsavgol_ref.so, gcc -O2, 1 thread) on a bounded sample of the same
    workload, timed on this host; falls back to the oracle port if the compiled reference did not travel."""
    import ctypes as C
    from oracle import sgo
    ref = os.path.join(ROOT, "oracle", "_ref", "libsavgol_ref.so")
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


def bench_stream(sg, a):
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
                       "roofline": {"bound": "hbm", "achieved": round(8.0 * samples / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(8.0 * samples / ms / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_sample": 8}},
        **({} if a.no_cpu else {"cpu_baseline": cpu_reference("stream")}),
    }))


def bench_image(sg, a):
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
                     "roofline": {"bound": "hbm", "achieved": round(8.0 * pix / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(8.0 * pix / ms / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_pixel": 8}}
    print(json.dumps({"workload": f"BASELINE config 4{'' if N == 512 and size == 4096 else ' (subset)'}: {N} images x {size}x{size} fp32, n=7, order 3, method {a.method}", "modes": res,
                      **({} if a.no_cpu else {"cpu_baseline": cpu_reference("image")})}))



def pmc_traffic(ch, length):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes of the same command
    (FETCH_SIZE x2 for gfx950 + WRITE_SIZE, separate passes; see profiles/*_pmc_summary.json).  bench.py cannot
    collect counters itself; the newest committed summary for the same workload is reported, else null."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_1d_f32_n32_pmc_summary.json"))):
        try:
            d = json.load(open(path))
            if d.get("algorithmic_bytes_per_launch") == 8.0 * ch * length:
                best = d["hbm_traffic_bytes_per_launch"]
        except Exception:
            pass
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--channels", type=int, default=4096, help="channels per GPU")
    ap.add_argument("--length", type=int, default=1 << 20)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--workload", choices=["batch1d", "batch1d_f64", "stream", "image"], default="batch1d",
                    help="batch1d = the headline (BASELINE config 2); batch1d_f64 = config 5's per-GPU shape (fp64, n=32, d=2, "
                         "2^22-sample channels, POLYNOMIAL; 1024 channels per GPU by default); stream / image = configs 3 / 4, "
                         "single GPU, extra JSON")
    ap.add_argument("--streams", type=int, default=65536)
    ap.add_argument("--ticks", type=int, default=4096)
    ap.add_argument("--images", type=int, default=512, help="2-D: frames per pass (BASELINE config 4 has 512 = 34 GB in + 34 GB out)")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--method", type=int, default=2, help="2-D: 1 = dense (bit-exact), 2 = separable")
    args = ap.parse_args()
    args.no_cpu = args.no_cpu
    if args.workload in ("stream", "image"):
        sg = load_package()
        (bench_stream if args.workload == "stream" else bench_image)(sg, args)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # test hooks (tests/test_gpu_bench_contract.py): run the N>1 plumbing on a 1-GPU box with every rank on one device
    backend = os.environ.get("SAVGOL_BENCH_BACKEND", "nccl")
    if "SAVGOL_BENCH_DEVICE" in os.environ:
        local = int(os.environ["SAVGOL_BENCH_DEVICE"])
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    sg = load_package()
    assert sg.lib().savgol_hip_set_device(local) == 0, sg.last_error()

    f64 = args.workload == "batch1d_f64"
    if f64:                                               # config 5 shape unless overridden on the command line
        if args.channels == 4096: args.channels = 1024
        if args.length == 1 << 20: args.length = 1 << 22
    ch, length = args.channels, args.length
    deriv = 2 if f64 else D
    modes = [0] if f64 else [0, 1, 2, 3]
    esize = 8 if f64 else 4
    x = torch.empty((ch, length), dtype=torch.float64 if f64 else torch.float32, device=dev)
    y = torch.empty_like(x)
    sg.synth(x, channel0=rank * ch)                       # generated in HBM, never crosses PCIe
    filters = [sg.Filter(N, M, deriv, 1.0, mode) for mode in modes]
    torch.cuda.synchronize()

    def step(events=None):
        for f in filters:
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            f.apply_batch(x, y, ch, length, dtype="f64" if f64 else "f32")
            if events is not None:
                e1.record(); events.append((e0, e1))

    for _ in range(args.warmup):
        step()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    events = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(events)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    if rank == 0:
        launches_ms = [a.elapsed_time(b) for a, b in events]
        avg_ms = float(np.mean(launches_ms))
        alg_bytes = 2.0 * esize * ch * length               # sizeof(T) read + sizeof(T) written per sample
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        samples = float(len(modes)) * ch * length * args.steps * world
        out = {
            "metric": "Msamples/s filtered (1D batch, hw=32, poly=4) + % HBM roofline" if not f64 else
                      "Msamples/s filtered (1D batch fp64, hw=32, poly=4, derivative=2) + % HBM roofline",
            "value": round(samples / elapsed / 1e6, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if f64 else "f32",
            "data": "synthetic",
            "config": {"workload": (f"BASELINE config 2: {ch} channels x {length} fp32 samples per GPU, half_window={N}, "
                                    f"poly_order={M}, derivative={D}, one pass per boundary mode "
                                    "(POLYNOMIAL, REFLECT, PERIODIC, CONSTANT) per step") if not f64 else
                                   (f"BASELINE config 5 shape: {ch} channels x {length} fp64 samples per GPU (the full config is 4096 per "
                                    f"GPU, processed in such chunks), half_window={N}, poly_order={M}, derivative=2, POLYNOMIAL"),
                       "channels_per_gpu": ch, "length": length, "sharding": "channels, no collective"},
            "roofline": {"bound": "hbm", "kernel": f"sg1d_center_kernel<{'double' if f64 else 'float'},{N}>",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None if f64 else pmc_traffic(ch, length),
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(avg_ms, 4),
                         "launches_timed": len(launches_ms)},
        }
        if world == 1 and not args.no_cpu:
            # CPU leg (rank 0, N=1 only): the reference timed on this host + a parity spot check of what was just
            # timed (last mode run = CONSTANT) against the CPU oracle -- the only place bench.py touches oracle/
            out["cpu_baseline"] = cpu_baseline(length, deriv)
            from oracle import sgo
            sample = [0, ch // 2, ch - 1]
            ref = sgo.Filter(N, M, deriv, 1.0, modes[-1]).apply_f64(x[sample].cpu().numpy().astype(np.float64))
            checked = float(np.max(np.abs(y[sample].cpu().numpy() - ref)) / np.max(np.abs(ref)))
            assert checked < (1e-12 if f64 else 1e-6), f"parity lost: normwise error {checked}"
            out["parity_normwise_vs_fp64_oracle"] = checked
        if world == 1 and not f64:
            # secondary figure, outside the timed region: the same four passes with the reference's own summation
            # order (SAVGOL_HIP_OPT_REFERENCE_SUMMATION -> outputs bit-identical to the reference library's)
            L = sg.lib()
            if L.savgol_hip_set_option(sg.SAVGOL_HIP_OPT_REFERENCE_SUMMATION, 1) == 0:
                try:
                    for flt in filters:
                        flt.apply_batch(x, y, ch, length)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for flt in filters:
                        flt.apply_batch(x, y, ch, length)
                    e1.record(); torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / len(filters)
                    out["bit_identical_mode"] = {"Msamples_per_s": round(ch * length / ms / 1e3, 1), "avg_launch_ms": round(ms, 4),
                                                 "roofline_frac": round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                 "note": "reference summation order (four chains, separate multiply and add): "
                                                         "outputs equal the reference library's bit for bit"}
                    if not args.no_cpu:
                        from oracle import sgo
                        sample = [0, ch - 1]
                        ref32 = sgo.Filter(N, M, deriv, 1.0, modes[-1]).apply(x[sample].cpu().numpy())
                        assert np.array_equal(y[sample].cpu().numpy().view(np.uint32), ref32.view(np.uint32)), "bit-identical mode lost"
                finally:
                    L.savgol_hip_set_option(sg.SAVGOL_HIP_OPT_REFERENCE_SUMMATION, 0)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
