#!/usr/bin/env python3
"""bench.py -- headline benchmark: Msamples/s filtered on the 1-D batch path (BASELINE.json config 2:
4096 channels x 2^20 fp32 samples per GPU, half_window=32, poly_order=4, all four boundary modes),
with the HBM-roofline fraction of the dominant kernel and the reference's CPU path timed beside it.

    python bench.py [--gpus N --steps K --warmup W]          N > 1: this process starts the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      (the driver's form)
    python bench.py --workload batch1d_f64 | stream | image [--rowband]      the other BASELINE configs as the main line

A step = one pass of savgol_apply_batch_f32 over the whole resident batch in EACH of the four boundary
modes (4 launches of the centre kernel; the POLYNOMIAL one carries its edge rows as extra items).  Inputs are generated in HBM
before the timed region.  Channels are independent: with N GPUs every rank owns its own 4096 channels
(weak scaling, no data-path collective); rank 0 prints one JSON line.  At N = 1 that line also carries, under
"extra", BASELINE configs 1, 3, 4 and the per-GPU slice of config 5, each with its own roofline and CPU baseline
(--no-extra skips them).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
N, M, D = 32, 4, 0
# the dominant kernels of configs 3 / 4 as rocprofv3 names them (profiles/r05_*_kernel_stats.csv)
STREAM_KERNEL_FMA = "sg_bank_dma_kernel<16,true,32,8,16,1,2,MomTaps<16,2>> (LDS-DMA tiles, 8-tick block moments; SAVGOL_HIP_STREAM_MOMENT=0: tap by tap; SAVGOL_HIP_STREAM_DMA=0: sg_bank_roll_kernel<16,true>)"
STREAM_KERNEL_REF = "sg_bank_dma_kernel<16,false,...> (SAVGOL_HIP_STREAM_DMA=0: sg_bank_roll_kernel<16,false>)"
IMAGE_KERNEL = "sg2d_rolling_kernel<7,2,1,true,false,20>"


# ---------------------------------------------------------------------------------------------------------------
# CPU legs: the reference's own code (oracle/_ref/libsavgol_ref.so, gcc -O2) on the host cores of this box
# ---------------------------------------------------------------------------------------------------------------
def _ref():
    """(ctypes lib of the compiled reference, make_golden helpers) or (None, None) when it did not travel."""
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libsavgol_ref.so")):
        from tests.golden import make_golden as mg
        return mg.load(), mg
    return None, None


def cpu_baseline(length, n=N, m=M, deriv=D, budget_s=12.0, all_cores=True):
    """The reference's savgol_apply (1 thread -- the reference has no threading) on a bounded sample of the same
    workload: as many `length`-sample channels as fit in ~budget_s."""
    import ctypes as C
    from oracle import sgo
    x = sgo.synth_f32(0, 1, length)[0]
    y = np.empty_like(x)
    L, mg = _ref()
    if L is not None:
        cfg = mg.Cfg(n, m, deriv, 1.0, 0)
        f = L.savgol_create(C.byref(cfg))
        run = lambda: L.savgol_apply(f, mg.fptr(x), mg.fptr(y), length)
        kind = "reference"
    else:
        flt = sgo.Filter(n, m, deriv)
        run = lambda: flt.apply(x)
        kind = "port"
    run()
    n_done, t0 = 0, time.perf_counter()
    while True:
        run(); n_done += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    out = {"value": round(n_done * length / el / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": kind,
           "sample": f"{n_done} channels x {length} fp32 samples (the reference is fp32 only; n={n}, m={m}, d={deriv}, POLYNOMIAL), "
                     f"savgol_apply back to back for {el:.1f} s, 1 thread"}
    # the same call from one thread per host core, each on its own channel (the reference has no threading
    # of its own; savgol_apply is re-entrant on a shared const filter and ctypes drops the GIL around it)
    if kind == "reference" and all_cores:
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


def cpu_reference(kind):
    """CPU leg of the stream / image workloads."""
    import ctypes as C
    from oracle import sgo
    L, mg = _ref()
    if kind == "stream":
        exe = os.path.join(ROOT, "oracle", "_ref", "cpu_stream_bench")
        if os.path.exists(exe):
            n_push = 20_000_000
            v = float(subprocess.run([exe, "16", "2", "1", "0.001", str(n_push)], capture_output=True, text=True, check=True).stdout)
            return {"kind": "reference", "cores": 1, "unit": "Msamples/s", "value": v, "ns_per_sample": round(1e3 / v, 2),
                    "sample": f"the reference's savgol_stream_push in a C loop, 1 stream, {n_push} samples, best of 5 (oracle/cpu_stream_bench.c)"}
        x = sgo.synth_f32(0, 1, 200000)[0]
        o = sgo.Stream(sgo.Filter(16, 2, 1, 1e-3))
        t0 = time.perf_counter()
        for v in x:
            o.push(v)
        el = time.perf_counter() - t0
        return {"kind": "port", "cores": 1, "unit": "Msamples/s", "value": round(200000 / el / 1e6, 3),
                "sample": "oracle push loop through ctypes (call-overhead bound)"}
    size = 1024
    img = sgo.synth_f32(0, size, size)
    out = np.zeros_like(img)
    res = {}
    for name, b in (("VALID", 0), ("CONSTANT", 1), ("REFLECT", 2)):
        if L is not None:
            cfg = mg.Cfg2(7, 7, 3, 0, 0, 1.0, 1.0)
            f = L.savgol2d_create(C.byref(cfg))
            t0 = time.perf_counter()
            L.savgol2d_apply(f, mg.fptr(img), size, size, size, mg.fptr(out), size, b)
            el = time.perf_counter() - t0
        else:
            f = sgo.Filter2D(7, 7, 3)
            t0 = time.perf_counter(); f.apply(img, size, b); el = time.perf_counter() - t0
        res[name] = round(size * size / el / 1e6, 2)
    return {"kind": "reference" if L is not None else "port", "cores": 1, "unit": "Mpix/s", "value": res,
            "sample": f"savgol2d_apply on one {size}x{size} fp32 frame, n=7, order 3, per boundary mode"}


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed(fn, reps=5, warm=1):
    """median HIP-event time of fn() in ms (torch's current stream is the stream every launch goes to)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = ev(), ev()
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def timed_back_to_back(fn, k=8, reps=3, warm=1):
    """ms per call of k calls enqueued in a row between one pair of events (median of reps) -- how the steps of a job follow each other, and
    how the headline's timed region is taken; timed() synchronises after every call, and a 1-2 ms kernel that starts on an idle chip pays the
    clock ramp every time (config 4 on 16 ... 64 frames: median 27-28 us per frame alone, 24.5 in a row or on 512 frames)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(k):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / k)
    return float(np.median(ts))


def sustain(fn, seconds=0.25):
    """keep the chip busy with fn() for `seconds` before a short kernel is timed: after a phase of sparse launches (config 3's per-tick latency loop:
    one 3 us kernel per 13 us) the memory and fabric clocks have stepped down and take tens of milliseconds of load to come back -- the same block
    push on the same buffers reads 0.67 / 0.57 (fused / bit-exact) in the first 20 ms and 0.72 / 0.64 from then on (tools/placement_stream_offset.py, R6.13)"""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(32):
            fn()
        torch.cuda.synchronize()


def roofline(alg_bytes, ms, launches_ms=None, **more):
    """`ms` = the average launch duration (what `achieved` is computed from); launches_ms = every timed launch, for the spread"""
    ach = alg_bytes / (ms * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
           "traffic": None, "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(ms, 4),
           # SURVEY 8(d)'s own quantity: the kernel's INPUT bytes alone (N x sizeof(T): half of read + write on every path here) over its time and 8 TB/s
           "kernel_read_only_frac": round(0.5 * ach / HBM_PEAK_GBS, 4), **more}
    if launches_ms is not None and len(launches_ms):
        med = float(np.median(launches_ms))
        out.update({"median_launch_ms": round(med, 4), "min_launch_ms": round(float(np.min(launches_ms)), 4),
                    "max_launch_ms": round(float(np.max(launches_ms)), 4), "frac_at_median": round(alg_bytes / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "launches_timed": len(launches_ms)})
    return out


def copy_ceiling(sg, x, y, reps=5):
    """what the memory system gives a PLAIN stream of the same two buffers, timed in this process with the same HIP events, before
    the timed region: a flat nontemporal copy x -> y (savgol_hip_stream_copy: one 16-byte vector per thread, the fastest copy shape measured,
    tools/membench2.hip) and a read of x alone (savgol_hip_stream_read).  Fractions are of 8 TB/s, bytes = read + written."""
    L = sg.lib()
    nbytes = x.numel() * x.element_size()
    # buffers a plain copy is through with in under ~2 ms (config 3's 1 GiB pair: 0.33 ms) are timed eight calls in a row, as the kernels on them are:
    # one short launch between two synchronises starts on an idle chip and pays its clock ramp (timed_back_to_back)
    t = timed_back_to_back if 2.0 * nbytes / 6.0e12 < 2.0e-3 else timed
    try:
        ms_c = t(lambda: L.savgol_hip_stream_copy(x.data_ptr(), y.data_ptr(), nbytes, None), reps=reps, warm=1)
        ms_r = t(lambda: L.savgol_hip_stream_read(x.data_ptr(), nbytes, None, None), reps=reps, warm=1)
    except AttributeError:
        return {}
    # read_ceiling_frac (round 5 called it read_only_frac): what a flat READ of the input buffer alone reaches -- a ceiling, not this kernel's figure
    # (that is roofline.kernel_read_only_frac).  savgol_hip_stream_copy / _read on the workload's own buffers, same process and clock, before the timed region.
    return {"copy_ms": round(ms_c, 4), "copy_frac": round(2.0 * nbytes / (ms_c * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "read_ceiling_ms": round(ms_r, 4), "read_ceiling_frac": round(nbytes / (ms_r * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


def add_ceiling(roof, ceil):
    """roofline.copy_frac / frac_of_copy / read_ceiling_frac"""
    if ceil and ceil.get("copy_frac"):
        roof.update({k: ceil[k] for k in ("copy_frac", "read_ceiling_frac", "copy_ms", "read_ceiling_ms")})
        roof["frac_of_copy"] = round(roof["frac"] / ceil["copy_frac"], 4)
    return roof


def normwise(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    den = float(np.max(np.abs(b)))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


def reference_fp32_1d(x32, n, m, d, dt, mode):
    """The REFERENCE's own fp32 output on these channels: the compiled reference (oracle/_ref/libsavgol_ref.so) when it travelled, else the
    oracle's restatement, which tests/test_oracle_pinned.py pins to it bit for bit.  Returns (outputs, which)."""
    import ctypes as C
    from oracle import sgo
    x32 = np.ascontiguousarray(x32, np.float32)
    L, mg = _ref()
    if L is None:
        return sgo.Filter(n, m, d, dt, mode).apply(x32), "oracle restatement (bit-pinned to the compiled reference)"
    cfg = mg.Cfg(n, m, d, dt, mode)
    f = L.savgol_create(C.byref(cfg))
    out = np.empty_like(x32)
    for c in range(x32.shape[0]):
        L.savgol_apply(f, mg.fptr(x32[c]), mg.fptr(out[c]), x32.shape[1])
    L.savgol_destroy(f)
    return out, "compiled reference (oracle/_ref/libsavgol_ref.so)"


def parity_fields(got, ref64, ref32, which):
    """what north_star names: the distance of the timed kernel's output from the reference's own (fp32) output -- beside its distance from
    the double-accumulation oracle and the reference's own distance from that oracle"""
    return {"parity_normwise_vs_fp64_oracle": normwise(got, ref64), "parity_normwise_vs_reference_fp32": normwise(got, ref32),
            "reference_fp32_own_error_vs_fp64_oracle": normwise(ref32, ref64), "reference_fp32_from": which}


def strip_notes(node, path="", sink=None):
    """The driver keeps the LAST 8 KB of the line (config 3 was cut off by ~25 KB of prose).  Explanatory `note` strings are
    taken out of the printed object; what each said is in profiles/README.md ("bench.py notes"), and a run under gpurun also leaves them in
    gpurun_out/bench_notes.json."""
    sink = {} if sink is None else sink
    if isinstance(node, dict):
        for k in list(node):
            if k == "note" and isinstance(node[k], str):
                sink[path or "."] = node.pop(k)
            else:
                strip_notes(node[k], f"{path}.{k}" if path else k, sink)
    elif isinstance(node, list):
        for i, v in enumerate(node):
            strip_notes(v, f"{path}[{i}]", sink)
    return sink


def _sig(v, digits=4):
    return None if v is None else float(f"{float(v):.{digits}g}")


def summary_of(out):
    """<= 1 KB at the very END of the line: every BASELINE config as {frac, ms, parity} (parity = normwise distance from the fp64 oracle, 0.0 = bit-identical
    to the reference), so that the tail the driver keeps carries all of them."""
    ex = out.get("extra", {})

    def get(node, *path):
        for k in path:
            if not isinstance(node, dict) or k not in node:
                return None
            node = node[k]
        return node

    def entry(frac, ms, parity, **more):
        e = {"frac": _sig(frac), "ms": _sig(ms), "parity": None if parity is None else float(f"{float(parity):.2e}")}
        e.update({k: v for k, v in more.items() if v is not None})
        return e
    S = {"c2": entry(get(out, "roofline", "frac"), get(out, "roofline", "avg_launch_ms"), out.get("parity_normwise_vs_fp64_oracle"),
                     spread_min=_sig(get(out, "roofline", "placement_spread", "frac_min")))}
    c1 = ex.get("config1", {})
    S["c1"] = entry(get(c1, "device_resident", "roofline", "frac"), get(c1, "device_resident", "ms"), c1.get("parity_normwise_vs_fp64_oracle"))
    bp = get(ex, "config3", "block_push") or {}
    S["c3_fused"] = entry(get(bp, "roofline", "frac"), bp.get("ms"), bp.get("parity_normwise_vs_fp64_oracle"),
                          spread_min=_sig(get(bp, "roofline", "placement_spread", "frac_min")), spread_med=_sig(get(bp, "roofline", "placement_spread", "frac_median")))
    br = get(ex, "config3", "block_push_reference_order") or {}
    S["c3_bit_exact"] = entry(br.get("roofline_frac"), br.get("ms"), 0.0 if str(br.get("parity", "")).startswith("bit-identical") else None,
                              spread_min=_sig(get(br, "placement_spread", "frac_min")), spread_med=_sig(get(br, "placement_spread", "frac_median")))
    for mode in ("VALID", "CONSTANT", "REFLECT"):
        m = get(ex, "config4", "modes", mode) or {}
        S["c4_" + mode] = entry(get(m, "roofline", "frac"), m.get("ms"), m.get("parity_normwise_vs_fp64_oracle"))
    rb = ex.get("config4_rowband", {})
    S["c4_rowband"] = entry(rb.get("roofline_frac_of_step"), rb.get("step_ms"), rb.get("parity_normwise_vs_fp64_oracle"))
    c5 = ex.get("config5_slice", {})
    S["c5"] = entry(get(c5, "roofline", "frac"), get(c5, "roofline", "avg_launch_ms"), c5.get("parity_normwise_vs_fp64_oracle"), kernel=get(c5, "roofline", "kernel"))
    for key, name in (("exact_1e12", "c5_exact_1e12"),):
        node = c5.get(key) or {}
        if node:
            S[name] = entry(get(node, "roofline", "frac"), get(node, "roofline", "avg_launch_ms"), node.get("parity_normwise_vs_fp64_oracle"))
    ip = c5.get("in_place") or {}
    S["c5_in_place"] = entry(ip.get("roofline_frac"), ip.get("ms_per_chunk"), ip.get("parity_normwise_vs_fp64_oracle"), over_pct=ip.get("over_out_of_place_pct"))
    S["push_wait_p50_us"] = get(ex, "config3", "from_c", "push_wait_us", "p50")
    errs = [k for k, v in ex.items() if isinstance(v, dict) and "error" in v]
    if errs:
        S["errors"] = errs
    return S


def emit(out):
    """print THE line: prose notes out, the compact summary last"""
    notes = strip_notes(out)
    if notes and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        try:
            with open(os.path.join(ROOT, "gpurun_out", "bench_notes.json"), "w") as fh:
                json.dump(notes, fh, indent=1)
        except OSError:
            pass
    if "extra" in out:
        out["summary"] = summary_of(out)
    try:
        # what C libraries still hold in their stdio buffers (RCCL's version banner: five lines it writes when the first communicator comes up) goes out
        # BEFORE the line, so that the JSON object is the last thing on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass
    print(json.dumps(out), flush=True)


def build_facts(sg):
    """facts about the binary this line was measured on that cannot be seen from inside the repository afterwards"""
    import glob
    import hashlib
    src = sorted(glob.glob(os.path.join(ROOT, "savitzky-golay-filter_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    newest = max(src, key=os.path.getmtime) if src else None
    lib = sg.LIB_PATH
    h = hashlib.sha256()
    with open(lib, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    out = {"library": os.path.relpath(lib, ROOT), "library_bytes": os.path.getsize(lib), "library_sha256_16": h.hexdigest()[:16],
           "library_mtime": int(os.path.getmtime(lib)), "newest_source": os.path.relpath(newest, ROOT) if newest else None,
           "newest_source_mtime": int(os.path.getmtime(newest)) if newest else None,
           "note": "mtimes are those of the snapshot this process ran from (a fresh copy may carry one mtime for every file); the sha identifies the binary"}
    out["library_not_older_than_sources"] = bool(newest is None or out["library_mtime"] >= out["newest_source_mtime"])
    try:
        out["kernel_source_sha"] = kernel_source_sha()
    except Exception:
        pass
    return out


def with_traffic(roof, pattern, files=None):
    """roofline.traffic from the committed PMC summary of this kernel, when it was taken on the sources this run is built from"""
    roof["traffic"], roof["traffic_source"] = pmc_traffic(roof["algorithmic_bytes_per_launch"], pattern, files)
    return roof


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 1: 1 channel x 1e6 samples, n=5, m=3, d=0, POLYNOMIAL -- the reference's own CPU-runnable case, and the
# reference's demo loop (test/iterative/test_savgol_main.c:136-155: 360 points, n=6, m=3, 10 000 iterations)
# ---------------------------------------------------------------------------------------------------------------
def bench_config1(sg, no_cpu):
    L1 = 1_000_000
    x = torch.empty((1, L1), dtype=torch.float32, device="cuda")
    sg.synth(x)
    y = torch.empty_like(x)
    f = sg.Filter(5, 3, 0, 1.0, 0)
    ms = timed(lambda: f.apply_batch(x, y, 1, L1), reps=20, warm=3)
    xh = x.cpu().numpy()[0]
    # the drop-in call as a C caller makes it: the same two host buffers call after call (a fresh numpy output per call would
    # time first-touch page faults under the D2H copy -- 14 ms instead of 0.2 -- not this library)
    yh = np.zeros_like(xh)
    for _ in range(3):
        f.apply(xh, out=yh)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); f.apply(xh, out=yh); ts.append(time.perf_counter() - t0)
    host_ms = float(np.median(ts)) * 1e3
    out = {"workload": "BASELINE config 1: 1 channel x 1e6 samples, half_window=5, poly_order=3, derivative=0, POLYNOMIAL (fp32: the "
                       "reference's API is fp32)",
           "device_resident": {"ms": round(ms, 4), "Msamples_per_s": round(L1 / ms / 1e3, 1),
                               "roofline": roofline(8.0 * L1, ms, kernel="sg1d_center_kernel<float,5>",
                                                    note="one 4 MB signal = 489 tiles on a 1024-SIMD chip: launch/latency bound, not a roofline case")},
           "host_pointer_savgol_apply": {"ms": round(host_ms, 3), "Msamples_per_s": round(L1 / host_ms / 1e3, 1),
                                         "note": "drop-in call on pageable host buffers reused across calls (median of 9): H2D + reference-order kernel + D2H, PCIe inclusive; bit-identical to the reference"}}
    if not no_cpu:
        out["cpu_baseline"] = cpu_baseline(L1, 5, 3, 0, budget_s=3.0, all_cores=False)
        from oracle import sgo
        ref = sgo.Filter(5, 3, 0, 1.0, 0).apply(xh)
        assert np.array_equal(yh.view(np.uint32), ref.view(np.uint32)), "config 1: host-pointer savgol_apply lost bit-identity"
        ref64 = sgo.Filter(5, 3, 0, 1.0, 0).apply_f64(xh.astype(np.float64)[None])[0]
        ref32, which = reference_fp32_1d(xh[None], 5, 3, 0, 1.0, 0)
        out.update(parity_fields(y.cpu().numpy()[0], ref64, ref32[0], which))       # the device-resident default kernel; the host call above is bit-identical
        assert out["parity_normwise_vs_fp64_oracle"] < 1e-6 and out["parity_normwise_vs_reference_fp32"] < 1e-6
        # the reference's demo loop (test_savgol_main.c:136-155: 360 points, n=6, m=3, 10 000 calls), on the compiled reference and on this
        # library in ONE C program with one clock (tools/time_demo360.c): the only source of the small-signal figures in README / INTEGRATION
        exe = os.path.join(ROOT, "savitzky-golay-filter_amd", "lib", "time_demo360")
        ref_so = os.path.join(ROOT, "oracle", "_ref", "libsavgol_ref.so")
        if os.path.exists(exe) and os.path.exists(ref_so):
            import subprocess
            r = subprocess.run([exe, ref_so, sg.LIB_PATH], capture_output=True, text=True, timeout=120)
            if r.returncode == 0 and r.stdout.strip().startswith("{"):
                both = json.loads(r.stdout.strip().splitlines()[-1])
                cpu, gpu = both[ref_so], both[sg.LIB_PATH]
                assert abs(cpu["out0"] - gpu["out0"]) < 1e-4, "the two libraries disagree on the demo's first output"
                out["reference_demo_360pt"] = {"program": "savitzky-golay-filter_amd/lib/time_demo360 (tools/time_demo360.c): dlopen()s both libraries, clock_gettime around 10 x 1000 savgol_apply calls",
                                               "cpu_reference": {k: cpu[k] for k in ("Msamples_per_s", "us_per_call_median")},
                                               "gpu_drop_in": {k: v for k, v in gpu.items() if k != "out0"},
                                               "note": "a 360-point signal is 1.4 KB: the reference's CPU loop finishes it in about a microsecond, the drop-in pays a "
                                                       "doorbell + completion round trip over PCIe per call (resident small-call service; a launched call costs ~20 us). "
                                                       "The GPU path is for batches and long signals; on tiny host signals the CPU reference is the faster library."}
    return out


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 3: 65 536 streams, n=16, m=2, d=1, dt=1e-3
# ---------------------------------------------------------------------------------------------------------------
def pin_to_gpu_numa_node():
    """Latency legs only: run on the CPUs of the socket the GPU hangs off (sysfs local_cpulist); a doorbell or a launch that
    crosses the socket interconnect first costs microseconds.  Returns the number of CPUs in the mask (0 = left alone)."""
    try:
        p = torch.cuda.get_device_properties(torch.cuda.current_device())
        bus = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        cpus = set()
        for part in open(f"/sys/bus/pci/devices/{bus}/local_cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        if cpus:
            os.sched_setaffinity(0, cpus)
        return len(cpus)
    except Exception:
        return 0


def bench_stream(sg, a):
    S, T, n = a.streams, a.ticks, 16
    pinned = pin_to_gpu_numa_node()
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
    # (c) block push: T ticks in one launch -- the fused-multiply-add bank (SAVGOL_STREAMBANK_FMA) and the reference-order one
    bank2 = sg.StreamBank(S, n, 2, 1, 1e-3, fma=True)
    # blocks follow each other in a stream job, and so they are timed: eight pushes in a row per measurement (the main line of --workload stream
    # times its K steps the same way); one 0.39 ms launch between two synchronises pays the clock ramp of an idle chip (3-6 %).  The copy
    # ceiling and the pushes are both taken on a chip that has been kept busy for a quarter of a second (sustain())
    sustain(lambda: bank2.push_block(x, T, out))
    ceil = copy_ceiling(sg, x, out)
    ms = timed_back_to_back(lambda: bank2.push_block(x, T, out), k=8, reps=5, warm=1)
    pick = [0, 1, S // 2, S - 1]
    got_fma = out[:, pick].cpu().numpy() if not a.no_cpu else None
    bank2r = sg.StreamBank(S, n, 2, 1, 1e-3)
    sustain(lambda: bank2r.push_block(x, T, out), 0.1)
    ms_ref = timed_back_to_back(lambda: bank2r.push_block(x, T, out), k=8, reps=5, warm=1)
    got_ref = out[:, pick].cpu().numpy() if not a.no_cpu else None
    samples = S * T
    res = {
        "workload": f"BASELINE config 3: {S} streams, n=16, m=2, d=1, dt=1e-3",
        "per_tick_launch": {"wall_latency_us_p50": round(float(lat[len(lat) // 2]), 2), "wall_latency_us_p99": round(float(lat[int(len(lat) * 0.99)]), 2),
                            "us_per_tick_back_to_back_from_python": round(tick_us, 3), "ns_per_sample": round(tick_us * 1e3 / S, 4),
                            "Msamples_per_s": round(S / tick_us, 1),
                            "note": "HIP events around 1000 pushes enqueued from Python: the interpreter's call rate, not the kernel (3.8 us in rocprofv3's "
                                    "kernel trace); from_c.us_per_tick_back_to_back is the same loop from plain C"},
        "block_push": {"ticks_per_launch": T, "ms": round(ms, 3), "ms_is": "per launch, eight launches in a row (median of 5)", "ns_per_sample": round(ms * 1e6 / samples, 5),
                       "Msamples_per_s": round(samples / ms / 1e3, 1),
                       "summation": "SAVGOL_STREAMBANK_FMA (fused multiply-adds; config 3's taps are linear in the tap index: whole 8-tick blocks of a window through two block moments, the <= 14 taps at its ends one by one; parity below)",
                       "roofline": add_ceiling(with_traffic(roofline(8.0 * samples, ms, kernel=STREAM_KERNEL_FMA, algorithmic_bytes_per_sample=8),
                                                            "r*_stream_block_pmc_summary.json", SOURCES_STREAM), ceil)},
        "block_push_reference_order": {"ms": round(ms_ref, 3), "Msamples_per_s": round(samples / ms_ref / 1e3, 1),
                                       "roofline_frac": round(8.0 * samples / (ms_ref * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "frac_of_copy": round(8.0 * samples / (ms_ref * 1e-3) / 1e9 / HBM_PEAK_GBS / ceil["copy_frac"], 4) if ceil.get("copy_frac") else None,
                                       "kernel": STREAM_KERNEL_REF,
                                       "note": "savgol_streambank_create: multiply and add rounded separately, bit-identical to the reference's savgol_stream_push"},
    }
    # PLACEMENT SPREAD: the same two launches on fresh pairs of buffers inside this process, the timed pair kept alive so that new
    # physical pages back the new ones -- the block push is 0.36-0.42 ms for ONE kernel depending on where its two 1 GiB buffers sit (R5.9)
    try:
        fr_f, fr_r = [], []
        keep = []
        for _ in range(3):
            free, _tot = torch.cuda.mem_get_info()
            if free < 3 * x.numel() * 4:
                break
            x2 = torch.empty_like(x); y2 = torch.empty_like(out)
            x2.copy_(x)
            keep += [x2, y2]
            bank2.push_block(x2, T, y2); bank2r.push_block(x2, T, y2); torch.cuda.synchronize()           # first touch of a fresh pair is slower whatever runs
            t_f = timed_back_to_back(lambda: bank2.push_block(x2, T, y2), k=8, reps=3, warm=1)
            t_r = timed_back_to_back(lambda: bank2r.push_block(x2, T, y2), k=8, reps=3, warm=1)
            fr_f.append(round(8.0 * samples / (t_f * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
            fr_r.append(round(8.0 * samples / (t_r * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
        del keep
        if fr_f:
            # frac_median: over the timed pair AND the fresh ones -- one placement is a lottery ticket (R5.9), the median is what a job sees
            res["block_push"]["roofline"]["placement_spread"] = {"fresh_pairs": len(fr_f), "frac_of_each": fr_f, "frac_min": min(fr_f), "frac_max": max(fr_f),
                                                                 "frac_median": round(float(np.median(fr_f + [res["block_push"]["roofline"]["frac"]])), 4)}
            res["block_push_reference_order"]["placement_spread"] = {"fresh_pairs": len(fr_r), "frac_of_each": fr_r, "frac_min": min(fr_r), "frac_max": max(fr_r),
                                                                     "frac_median": round(float(np.median(fr_r + [res["block_push_reference_order"]["roofline_frac"]])), 4)}
    except Exception as e:                                               # noqa: BLE001 -- diagnostic only
        res["block_push"]["roofline"]["placement_spread"] = {"error": f"{type(e).__name__}: {e}"}
    if not a.no_cpu:
        # parity of what was just timed, on four streams: the reference's own stream arithmetic (one chain, product and sum rounded: the oracle's
        # push loop, pinned bit for bit to the compiled reference) and the double-accumulation oracle
        from oracle import sgo
        xh = x[:, pick].cpu().numpy()
        flt = sgo.Filter(n, 2, 1, 1e-3)
        ref64 = flt.apply_f64(xh.T.astype(np.float64).copy())[:, n:T - n]
        seq = np.stack([np.array([v for v, ok in (o.push(v) for v in xh[:, j]) if ok], np.float32) for j, o in ((j, sgo.Stream(flt)) for j in range(len(pick)))])
        assert np.array_equal(got_ref[2 * n:].T.view(np.uint32), seq.view(np.uint32)), "config 3: the reference-order bank lost bit-identity"
        res["block_push"].update(parity_fields(got_fma[2 * n:].T, ref64, seq, "oracle push loop (bit-pinned to the compiled reference's savgol_stream_push)"))
        res["block_push_reference_order"]["parity"] = "bit-identical to the reference's push loop on 4 sampled streams x %d ticks" % (T - 2 * n)
        bar = max(1e-6, 1.1 * res["block_push"]["reference_fp32_own_error_vs_fp64_oracle"])
        assert res["block_push"]["parity_normwise_vs_fp64_oracle"] < bar, res["block_push"]
    # (e) the resident tick service: a doorbell + a completion array per tick instead of a launch + synchronise
    torch.cuda.synchronize()
    try:
        bank3 = sg.StreamBank(S, n, 2, 1, 1e-3)
        bank3.service_start(2000)
        for t in range(100):
            bank3.service_tick(x[t % T], o1)
        lat2 = []
        for t in range(100, 100 + 4000):
            t0 = time.perf_counter()
            rc = bank3.service_tick(x[t % T], o1)
            lat2.append((time.perf_counter() - t0) * 1e6)
            assert rc == 1, sg.last_error()
        bank3.service_stop()
        lat2 = np.sort(np.array(lat2))
        res["resident_service"] = {"wall_latency_us_p50": round(float(lat2[len(lat2) // 2]), 2), "wall_latency_us_p99": round(float(lat2[int(len(lat2) * 0.99)]), 2),
                                   "note": "savgol_streambank_service_tick through the Python binding (ctypes adds ~2 us per call; examples/c_api_demo.c measures it "
                                           "from C); outputs bit-identical to the per-tick kernel (tests/test_gpu_stream.py)"}
    except Exception as e:
        res["resident_service"] = {"error": f"{type(e).__name__}: {e}"}
    res["cpus_pinned_to_gpu_numa_node"] = pinned
    # the same two per-tick paths timed from plain C (examples/c_api_demo.c, built by `make` next to the library): no
    # interpreter between the doorbell and the completion array
    demo = os.path.join(ROOT, "savitzky-golay-filter_amd", "lib", "c_api_demo")
    if os.path.exists(demo):
        try:
            import re
            txt = subprocess.run([demo], capture_output=True, text=True, timeout=120).stdout
            m1 = re.search(r"p50 ([0-9.]+) us\s+p99 ([0-9.]+) us\s+\(launch \+ sync", txt)
            m2 = re.search(r"p50 ([0-9.]+) us\s+p99 ([0-9.]+) us\s+\(resident service", txt)
            m3 = re.search(r"([0-9.]+) us per tick back to back", txt)
            m4 = re.search(r"p50 ([0-9.]+) us\s+p99 ([0-9.]+) us\s+\(push_wait", txt)
            if m1 and m2 and "c_api_demo: OK" in txt:
                res["from_c"] = {"launch_plus_synchronise_us": {"p50": float(m1.group(1)), "p99": float(m1.group(2))},
                                 "us_per_tick_back_to_back": float(m3.group(1)) if m3 else None,
                                 "push_wait_us": {"p50": float(m4.group(1)), "p99": float(m4.group(2))} if m4 else None,
                                 "resident_service_us": {"p50": float(m2.group(1)), "p99": float(m2.group(2))},
                                 "note": "examples/c_api_demo.c: 2000 ticks each, thread pinned to the GPU's NUMA node; the service's outputs are compared bit "
                                         "for bit with the per-tick kernel on a twin bank inside the program"}
        except Exception as e:
            res["from_c"] = {"error": f"{type(e).__name__}: {e}"}
    # (d) the single-stream drop-in call: one sample per savgol_stream_push (launch + sync per sample)
    s1 = sg.Stream(n, 2, 1, 1e-3)
    xs = x[:, 0].cpu().numpy()
    for v in xs[:64]:
        s1.push(float(v))
    t0 = time.perf_counter()
    for v in xs[64:64 + 500]:
        s1.push(float(v))
    res["single_stream_push"] = {"us_per_sample": round((time.perf_counter() - t0) / 500 * 1e6, 2),
                                 "note": "savgol_stream_push on one SavgolStream: one doorbell round trip to the resident small-call service per sample "
                                         "(rounds 1-2: a launch and a synchronise, ~18 us); the reference's CPU push costs ~50 ns (cpu_baseline below) -- "
                                         "one stream at a time is the CPU's case, the bank API is the GPU's"}
    if not a.no_cpu:
        res["cpu_baseline"] = cpu_reference("stream")
    return res


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 4: 512 images x 4096 x 4096 fp32, n=7, order 3
# ---------------------------------------------------------------------------------------------------------------
def bench_image(sg, a):
    Nimg, size, n = a.images, a.size, 7
    x = torch.empty((Nimg * size, size), dtype=torch.float32, device="cuda")
    sg.synth(x)
    y = torch.empty_like(x)
    f = sg.Filter2D(n, n, 3)
    res = {}
    ceil = copy_ceiling(sg, x, y, reps=3)
    crop = None
    for name, b in (("VALID", 0), ("CONSTANT", 1), ("REFLECT", 2)):
        ms = timed(lambda: f.apply_batch(x, y, size, size, Nimg, boundary=b, method=a.method), reps=5, warm=1)
        pix = Nimg * size * size
        res[name] = {"ms": round(ms, 3), "Mpix_per_s": round(pix / ms / 1e3, 1),
                     "roofline": add_ceiling(roofline(8.0 * pix, ms, algorithmic_bytes_per_pixel=8, kernel=IMAGE_KERNEL if a.method == 2 else "sg2d_dense_roll_kernel<7>"), ceil)}
        if not a.no_cpu and size >= 512:
            # parity of what was just timed on a 256 x 256 crop of the last frame's top-left corner (frame edges included): against the
            # reference's own dense fp32 sum (savgol2d_apply: compiled reference or its bit-pinned restatement) and the double oracle
            from oracle import sgo
            k0 = (Nimg - 1) * size
            sub = x[k0:k0 + 256 + n, :256 + n].cpu().numpy()                       # n extra rows / columns so that the crop's inner edge is exact
            o2 = sgo.Filter2D(n, n, 3)
            lo = n if b == 0 else 0
            got = y[k0 + lo:k0 + 256, lo:256].cpu().numpy()
            hi64 = o2.apply_f64acc(sub, sub.shape[1], b if b else 1)[lo:256, lo:256]
            ref32 = o2.apply(sub, sub.shape[1], b if b else 1)[lo:256, lo:256]
            res[name].update(parity_fields(got, hi64, ref32, "oracle restatement of savgol2d_apply (bit-pinned to the compiled reference)"))
            assert res[name]["parity_normwise_vs_fp64_oracle"] < 1e-6, (name, res[name])
        if a.method == 2:                                # the rolling kernel's committed counter passes, if taken on these sources
            traffic, src = pmc_traffic(8.0 * pix, "r*_2d_config4_pmc_summary.json", SOURCES_2D)
            res[name]["roofline"]["traffic"] = traffic
            res[name]["roofline"]["traffic_source"] = src
    out = {"workload": f"BASELINE config 4{'' if Nimg == 512 and size == 4096 else ' (subset)'}: {Nimg} images x {size}x{size} fp32, n=7, order 3, "
                       f"method {a.method} ({'dense, bit-identical' if a.method == 1 else 'exact low-rank row+column passes'})",
           "modes": res}
    if not a.no_cpu:
        out["cpu_baseline"] = cpu_reference("image")
    del x, y
    torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 4's row-band split on ONE GPU: the C exchange (savgol2d_rowband_exchange_rccl_peers on a one-rank
# communicator: a ring of one -- the frame is vertically periodic, the halo above the band is its own last ny rows) on its own stream, the band
# kernel meanwhile, then the edge strips: every call of the multi-GPU step runs with real bytes, and its parts are timed apart and together.
# ---------------------------------------------------------------------------------------------------------------
def bench_rowband_ring_of_one(sg, a):
    import importlib
    rccl = importlib.import_module("savgol_amd.rccl")
    if not rccl.available():
        return {"skipped": "librccl.so / lib/libsavgol_hip_rccl.so did not load"}
    Nimg, size, n = min(a.images, 64), a.size, 7
    L = sg.lib()
    comm = rccl.Comm(1, 0, rccl.unique_id())
    try:
        x = torch.empty((Nimg * size, size), dtype=torch.float32, device="cuda")
        sg.synth(x)
        band = x.view(Nimg, size, size)
        out = torch.empty_like(band)
        up = torch.empty((Nimg, n, size), dtype=torch.float32, device="cuda")
        dn = torch.empty_like(up)
        scratch = torch.empty((2, Nimg, n, size), dtype=torch.float32, device="cuda")
        f2 = sg.Filter2D(n, n, 3)
        main = torch.cuda.current_stream()
        # (normal priority: a high-priority side stream gets the exchange through in 45 us but costs the band 0.38 ms -- EXPERIMENTS R6.11)
        xs = torch.cuda.Stream()

        def exchange(stream):
            comm.rowband_exchange(band, n, up, dn, scratch, peers=(0, 0), stream=stream)

        def band_kernel(first=0, count=None):
            count = Nimg - first if count is None else count
            assert L.savgol2d_apply_batch_f32(f2.ptr, band[first].data_ptr(), size, size, size, size * size, out[first].data_ptr(), size, size * size, count, 1, a.method,
                                              None) == 0, sg.last_error()

        def edges():
            assert L.savgol2d_apply_rowband_edges_f32(f2.ptr, band.data_ptr(), size, size, size, size * size, up.data_ptr(), dn.data_ptr(), size, n * size,
                                                      out.data_ptr(), size, size * size, Nimg, 1, a.method, None) == 0, sg.last_error()

        def step():
            # exchange, then the strips' gather + filter, on the side stream; the band on the main one; only the copy of the strips' finished
            # rows is ordered behind the band (savgol2d_apply_rowband_edges_streams_f32 records that event itself)
            ready = torch.cuda.Event(); ready.record(main)
            xs.wait_event(ready)
            exchange(xs)
            if HEAD:
                # the band in two launches: a short head, then the rest.  RCCL's send / recv kernel cannot find registers beside a band launch that fills
                # every wave slot (R6.11); it gets them in the gap between the two launches and then runs -- and finishes -- beside the second one
                band_kernel(0, HEAD); band_kernel(HEAD)
            else:
                band_kernel()
            assert L.savgol2d_apply_rowband_edges_streams_f32(f2.ptr, band.data_ptr(), size, size, size, size * size, up.data_ptr(), dn.data_ptr(), size, n * size,
                                                              out.data_ptr(), size, size * size, Nimg, 1, a.method, xs.cuda_stream, None) == 0, sg.last_error()
        HEAD = int(os.environ.get("SAVGOL_BENCH_ROWBAND_HEAD", "4")) if Nimg >= 16 else 0
        ms_x = timed(lambda: exchange(main), reps=7, warm=2)
        ms_b = timed(band_kernel, reps=5, warm=1)
        ms_e = timed(edges, reps=7, warm=2)
        ms_step_alone = timed(step, reps=5, warm=1)
        ms_b_row = timed_back_to_back(band_kernel)
        ms_step = timed_back_to_back(step)               # steps as a job issues them: one after the other
        torch.cuda.synchronize()
        res = {"workload": f"BASELINE config 4 shape as ONE row band: {Nimg} frames x {size}x{size} fp32, n=7, order 3, CONSTANT left / right, the {n}-row halos "
                           "through savgol2d_rowband_exchange_rccl_peers on a one-rank communicator (ring of one: vertically periodic frames)",
               "rccl_ranks": comm.count(), "exchange": "savgol2d_rowband_exchange_rccl_peers (C ABI: pack launch + ncclSend / ncclRecv per side, own stream)",
               "edge_strips": "savgol2d_apply_rowband_edges_streams_f32: both strips of every frame as one batch, gathered and filtered on the exchange's stream beside "
                              "the band; only the copy of their finished rows waits for the band",
               "exchange_ms": round(ms_x, 4), "halo_bytes_per_side": Nimg * n * size * 4, "band_ms": round(ms_b, 4), "edge_strips_ms": round(ms_e, 4),
               "step_ms": round(ms_step, 4), "step_ms_is": "8 steps enqueued in a row / 8 (the parts above and step_alone_ms: one call, then a synchronise)",
               "band_in_a_row_ms": round(ms_b_row, 4), "step_alone_ms": round(ms_step_alone, 4), "serial_sum_ms": round(ms_x + ms_b + ms_e, 4),
               "overlap_ms": round(ms_x + ms_b + ms_e - ms_step_alone, 4),
               "exchange_hidden_frac": round(max(0.0, min(1.0, (ms_x + ms_b + ms_e - ms_step_alone) / ms_x)), 3) if ms_x > 0 else None,
               "exchange_and_strips_hidden_frac": round(max(0.0, min(1.0, (ms_x + ms_b + ms_e - ms_step_alone) / (ms_x + ms_e))), 3) if ms_x + ms_e > 0 else None,
               "Mpix_per_s": round(Nimg * size * size / ms_step / 1e3, 1),
               "roofline_frac_of_step": round(8.0 * Nimg * size * size / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if not a.no_cpu:
            # the step's output on the periodic extension: top rows of the last frame (they saw the halo from the bottom of the same frame)
            from oracle import sgo
            k = Nimg - 1
            xh = band[k].cpu().numpy()
            ext = np.concatenate([xh[size - n:], xh[:64 + n]], axis=0)[:, :256 + n]
            o2 = sgo.Filter2D(n, n, 3)
            want64 = o2.apply_f64acc(ext, ext.shape[1], 1)[n:n + 64, :256]
            ref32 = o2.apply(ext, ext.shape[1], 1)[n:n + 64, :256]
            got = out[k, :64, :256].cpu().numpy()
            res.update(parity_fields(got, want64, ref32, "oracle restatement of savgol2d_apply on the periodic extension"))
            assert res["parity_normwise_vs_fp64_oracle"] < 1e-6, res
        return res
    finally:
        comm.close()


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 5, per-GPU slice: 4096 channels x 2^22 fp64 (137 GB), n=32, m=4, d=2, POLYNOMIAL, in 1024-channel chunks
# ---------------------------------------------------------------------------------------------------------------
def config5_buffers(sg, channels, chunk, length, rank, dev):
    """Input slice resident in HBM (generated there, chunk by chunk); output for the whole slice when it fits beside it,
    else one chunk-sized output buffer that every chunk overwrites."""
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info(dev)
    row = length * 8
    margin = 6 << 30
    resident = min(channels, max(chunk, int((free - margin - chunk * row) // row) // chunk * chunk))
    x = torch.empty((resident, length), dtype=torch.float64, device=dev)
    for c0 in range(0, resident, chunk):
        sg.synth(x[c0:c0 + chunk], channel0=rank * channels + c0)
    free, _ = torch.cuda.mem_get_info(dev)
    full_out = free - margin >= resident * row
    y = torch.empty((resident if full_out else chunk, length), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    return x, y, resident, full_out


C5_TOL = 1e-6            # BASELINE config 5 / north_star: "fp64 output within 1e-6 of the reference" -- stated in the call (savgol_apply_batch_f64_tol)


def bench_config5_slice(sg, a, rank=0, dev=None, steps=2):
    """config 5's per-GPU slice through savgol_apply_batch_f64_tol(rel_tol = 1e-6): the tolerance the config states picks the kernel (round 6, VERDICT r05
    next #6).  `roofline` is THAT call, its parity asserted against the bar it was given; `exact_1e12` is the default tap-by-tap path on the same buffers."""
    dev = dev or torch.device("cuda", torch.cuda.current_device())
    channels, chunk, length = a.c5_channels, a.c5_chunk, 1 << 22
    x, y, resident, full_out = config5_buffers(sg, channels, chunk, length, rank, dev)
    f = sg.Filter(N, M, 2, 1.0, 0)
    ceil = copy_ceiling(sg, x[:chunk], y[:chunk], reps=3)              # one chunk's bytes = one launch's bytes
    alg = 16.0 * chunk * length
    c0_last = resident - chunk
    sample = [0, chunk - 1]
    ref = None
    if not a.no_cpu:
        from oracle import sgo
        ref = sgo.Filter(N, M, 2, 1.0, 0).apply_f64(x[c0_last:c0_last + chunk][sample].cpu().numpy())

    def leg(kw, kernel, pmc, bar):
        per_launch = []

        def one_pass():
            for c0 in range(0, resident, chunk):
                e0, e1 = ev(), ev()
                e0.record()
                f.apply_batch(x[c0:c0 + chunk], y[c0:c0 + chunk] if full_out else y, chunk, length, dtype="f64", **kw)
                e1.record(); per_launch.append((e0, e1))
        one_pass(); torch.cuda.synchronize(); per_launch.clear()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_pass()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        lms = [p.elapsed_time(q) for p, q in per_launch]
        node = {"call": "savgol_apply_batch_f64_tol(rel_tol=%g)" % kw["rel_tol"] if "rel_tol" in kw else "savgol_apply_batch_f64_ex(flags=0)",
                "Msamples_per_s": round(resident * length * steps / el / 1e6, 1), "ms_per_pass": round(el / steps * 1e3, 3),
                "roofline": add_ceiling(with_traffic(roofline(alg, float(np.mean(lms)), lms, kernel=kernel), pmc), ceil)}
        if ref is not None:
            got = (y[c0_last:c0_last + chunk] if full_out else y)[sample].cpu().numpy()
            node["parity_normwise_vs_fp64_oracle"] = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
            assert node["parity_normwise_vs_fp64_oracle"] < bar, f"config 5 parity lost: {node['parity_normwise_vs_fp64_oracle']} (bar {bar})"
        return node

    out = {"workload": f"BASELINE config 5, one GPU's slice: {resident} of {channels} channels x {length} fp64 samples resident in HBM "
                       f"({resident * length * 8 / 1e9:.1f} GB in), half_window=32, poly_order=4 (BASELINE names no order; d=2 needs >= 2), "
                       f"derivative=2, POLYNOMIAL, processed in {chunk}-channel chunks; "
                       + ("output slice resident too" if full_out else "every chunk writes the same chunk-sized output buffer (the slice's output does not fit beside its input)"),
           "rel_tol": C5_TOL}
    out.update(leg({"rel_tol": C5_TOL}, "sg1d_center_moment64_kernel<32,5>", "r*_1d_f64m_n32_pmc_summary.json", C5_TOL))
    if not a.no_cpu:
        out["cpu_baseline"] = {"note": "the reference has no fp64 path (SURVEY.md fact 1); its fp32 savgol_apply at this shape is the headline's cpu_baseline "
                                       "(same 65-tap loop, n=32)"}
    try:
        out["exact_1e12"] = leg({"flags": 0}, "sg1d_center_kernel<double,32>", "r*_1d_f64_n32_pmc_summary.json", 1e-12)
    except Exception as e:                                               # noqa: BLE001 -- a secondary leg must not take the primary figure down
        out["exact_1e12"] = {"error": f"{type(e).__name__}: {e}"}
    # IN PLACE (savgol_apply_batch_f64* with d_out == d_in): one chunk, on a copy of its input that sits in the output slice -- what lets config 5's
    # 137 GB slice run with ONE resident buffer.  Timed per call (every launch of it), the same arithmetic as the primary figure and as exact_1e12.
    try:
        if full_out:
            buf = y[c0_last:c0_last + chunk]
            for name, kw, bar, base in (("in_place", {"rel_tol": C5_TOL}, C5_TOL, out), ("in_place_exact_1e12", {"flags": 0}, 1e-12, out.get("exact_1e12", {}))):
                times = []
                for i in range(4):
                    buf.copy_(x[c0_last:c0_last + chunk])
                    e0, e1 = ev(), ev()
                    e0.record()
                    f.apply_batch(buf, buf, chunk, length, dtype="f64", **kw)
                    e1.record(); torch.cuda.synchronize()
                    if i:
                        times.append(e0.elapsed_time(e1))
                t = float(np.mean(times))
                inp = {"ms_per_chunk": round(t, 3), "roofline_frac": round(alg / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                oop = (base.get("roofline") or {}).get("avg_launch_ms")
                if oop:
                    inp["over_out_of_place_pct"] = round(100.0 * (t / oop - 1.0), 1)
                if ref is not None:
                    inp["parity_normwise_vs_fp64_oracle"] = float(np.max(np.abs(buf[sample].cpu().numpy() - ref)) / np.max(np.abs(ref)))
                    assert inp["parity_normwise_vs_fp64_oracle"] < bar, inp
                out[name] = inp
    except Exception as e:                                               # noqa: BLE001
        out["in_place"] = {"error": f"{type(e).__name__}: {e}"}
    del x, y
    torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------------------
def kernel_source_sha(files=None):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_summary import kernel_source_sha as sha
    return sha(files)


SOURCES_2D = ["sg_2d_roll.hip", "sg_2d.hpp", "sg_2d.hip"]            # what profiles/r*_2d_config4_pmc_summary.json is stamped with
SOURCES_STREAM = ["sg_stream_dma.hip", "sg_stream_roll.hip", "sg_stream_roll.hpp", "sg_stream_host.hpp", "sg_stream_moment_fit.cpp", "sg_stream.hpp", "sg_pk.hpp"]   # ... r*_stream_block_pmc_summary.json


def pmc_traffic(alg_bytes, pattern="r*_1d_f32_n32_pmc_summary.json", files=None):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this command (FETCH_SIZE x2
    for gfx950 + WRITE_SIZE, separate passes: profiles/*_pmc_summary.json, written by tools/pmc_summary.py).  bench.py cannot
    collect counters itself, so it reports a committed summary ONLY when that summary was taken on the kernel sources
    this run is built from (sha256 of csrc/sg_k1d*, sg_pk.hpp and sg_api_1d.cpp, recorded in the summary) -- otherwise null."""
    import glob
    try:
        cur = kernel_source_sha(files)
    except Exception:
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("kernel_source_sha") == cur and d.get("algorithmic_bytes_per_launch") == alg_bytes and "hbm_traffic_bytes_per_launch" in d:
            return d["hbm_traffic_bytes_per_launch"], {"file": os.path.relpath(path, ROOT), "kernel_source_sha": cur}
    return None, {"kernel_source_sha": cur, "note": "no committed PMC summary matches these kernel sources"}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class Run:
    """what every workload of one bench.py process shares: the arguments, this rank's place in the job, the library, the timed region"""

    def __init__(self, args):
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}: launch as `python bench.py --gpus N` or give torchrun the same N")
        self.dist = None
        # test hooks (tests/): run the N>1 plumbing with every rank on one device / without a device at all
        self.backend = os.environ.get("SAVGOL_BENCH_BACKEND", "nccl")
        self.dry = os.environ.get("SAVGOL_BENCH_DRYRUN") == "1"
        if "SAVGOL_BENCH_DEVICE" in os.environ:
            self.local = int(os.environ["SAVGOL_BENCH_DEVICE"])
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl" and not self.dry:
                torch.cuda.set_device(self.local)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local))
            else:
                dist.init_process_group("gloo")
        self.common = {"n_gpus": self.world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
                       "vs_baseline": None, "data": "synthetic"}
        self.dev = None
        self.sg = None

    def attach_gpu(self):
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)
        self.sg = load_package()
        assert self.sg.lib().savgol_hip_set_device(self.local) == 0, self.sg.last_error()

    def barrier(self):
        if self.dev is not None:
            torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        if self.dev is not None:
            torch.cuda.synchronize()

    def timed_region(self, step):
        """W untimed steps, then K steps between barriers; returns (max-over-ranks seconds, the per-launch event pairs)"""
        for _ in range(self.args.warmup):
            step(None)
        events = []
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(self.args.steps):
            step(events)
        self.barrier()
        el = time.perf_counter() - t0
        self.per_rank = None
        if self.dist is not None:
            dev = self.dev if self.backend == "nccl" else "cpu"
            # every rank's own wall time per step and mean launch duration (its HIP events), gathered: a SCALE line can then be read against the N = 1 line
            # rank by rank (VERDICT r05 next #9b) -- the job's figure stays the MAX over ranks
            launch = float(np.mean([a.elapsed_time(b) for a, b in events])) if events else 0.0
            mine = torch.tensor([el / self.args.steps * 1e3, launch], dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(mine) for _ in range(self.world)]
            self.dist.all_gather(allr, mine)
            self.per_rank = {"ms_per_step": [round(float(v[0]), 4) for v in allr], "avg_launch_ms": [round(float(v[1]), 4) for v in allr]}
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            el = t.item()
        return el, events

    def per_rank_fields(self, alg_bytes_per_launch):
        """N > 1: per-rank timings and each rank's own roofline fraction (the N = 1 line's `roofline.frac`, rank by rank), the communicator's size"""
        if self.per_rank is None:
            return {}
        ms = self.per_rank["ms_per_step"]
        fr = [round(alg_bytes_per_launch / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if v > 0 else None for v in self.per_rank["avg_launch_ms"]]
        return {"per_rank": {"ms_per_step_min": min(ms), "ms_per_step_max": max(ms), "ms_per_step_each": ms, "avg_launch_ms_each": self.per_rank["avg_launch_ms"],
                             "roofline_frac_each": fr},
                "backend": self.backend, "rccl_ranks": self.dist.get_world_size() if self.backend == "nccl" else None}

    def agree(self, ok):
        """every rank learns whether EVERY rank is fine (MIN all-reduce): a rank must never leave the others inside a collective"""
        if self.dist is None:
            return bool(ok)
        t = torch.tensor([1.0 if ok else 0.0], device=self.dev if (self.backend == "nccl" and self.dev is not None) else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return t.item() == 1.0

    def finish(self, code=0):
        if self.dist is not None:
            self.dist.destroy_process_group()
        if code:
            sys.exit(code)


def dry_run(r, args):
    """no GPU: the launch / barrier / max-over-ranks plumbing, and -- for --rowband -- the band plan and the choice of exchange
    (tests/test_bench_selflaunch.py)"""
    t = torch.tensor([1.0 + r.rank], dtype=torch.float64)
    if r.dist is not None:
        r.dist.barrier(); r.dist.all_reduce(t, op=r.dist.ReduceOp.MAX)
    line = {"metric": "dry run (no GPU work)", "n_gpus": r.world, "max_over_ranks": t.item(), "workload": args.workload}
    if args.workload == "image" and args.rowband:
        import importlib
        load_package()                                            # registers the package under its importable name; touches no GPU
        rowband = importlib.import_module("savgol_amd.rowband")
        band = rowband.RowBand(args.size * r.world, 7, r.rank, r.world)
        rows = torch.tensor([band.hi - band.lo], dtype=torch.float64)
        if r.dist is not None:
            r.dist.all_reduce(rows, op=r.dist.ReduceOp.SUM)
        line.update({"rowband": True, "exchange": args.exchange, "rows_over_ranks": int(rows.item()), "frame_rows": args.size * r.world,
                     "rank0_band": [band.lo, band.hi], "rank0_neighbours": {"up": bool(band.top), "down": bool(band.bottom)}})
    if r.rank == 0:
        print(json.dumps(line), flush=True)
    r.finish()


# ---------------------------------------------------------------------------------------------------------------
# workloads as the MAIN line
# ---------------------------------------------------------------------------------------------------------------
def run_stream(r):
    """BASELINE config 3 as the main line: every GPU its own 65 536 streams (weak scaling, no collective), one block push per step"""
    args, sg, dev = r.args, r.sg, r.dev
    lo, hi = sg.shard_range(args.streams * r.world, r.world, r.rank)
    S, T = hi - lo, args.ticks
    x = torch.empty((T, S), dtype=torch.float32, device=dev); sg.synth(x, channel0=lo)
    y = torch.empty_like(x)
    bank = sg.StreamBank(S, 16, 2, 1, 1e-3, fma=True)       # SAVGOL_STREAMBANK_FMA; the reference-order bank is timed beside it under "latency"
    # a step is 0.39 ms: W warm-up steps do not bring the memory clocks back up after the set-up phase; a quarter of a second of the same pushes does (R6.13)
    sustain(lambda: bank.push_block(x, T, y))
    ceil = copy_ceiling(sg, x, y)

    def step(events):
        e0, e1 = ev(), ev()
        e0.record(); bank.push_block(x, T, y); e1.record()
        if events is not None:
            events.append((e0, e1))
    el, events = r.timed_region(step)
    if r.rank == 0:
        lms = [p.elapsed_time(q) for p, q in events]
        ms = float(np.mean(lms))
        out = {"metric": "Msamples/s filtered (streaming, 65536 streams per GPU, hw=16, poly=2, derivative=1, block push)",
               "value": round(S * T * args.steps * r.world / el / 1e6, 1), "unit": "Msamples/s", **r.common,
               "ms_per_step": round(el / args.steps * 1e3, 4), "dtype": "f32",
               "config": {"workload": f"BASELINE config 3: {S} streams per GPU x {T} ticks per step, n=16, m=2, d=1, dt=1e-3", "sharding": "streams, no collective"},
               "roofline": add_ceiling(with_traffic(roofline(8.0 * S * T, ms, lms, kernel=STREAM_KERNEL_FMA), "r*_stream_block_pmc_summary.json", SOURCES_STREAM), ceil)}
        out.update(r.per_rank_fields(8.0 * S * T))
        if r.world == 1 and not args.no_extra:
            out["latency"] = bench_stream(sg, args)
            if "cpu_baseline" in out["latency"]:
                out["cpu_baseline"] = out["latency"]["cpu_baseline"]
        emit(out)
    r.finish()


def rowband_comm(r, exchange):
    """The C exchange's communicator for --rowband at N > 1, or None for --exchange torch.  --exchange c that cannot come up ends the job
    with a non-zero exit code on EVERY rank (round 4 fell back to torch.distributed with only a string in the line)."""
    if exchange != "c":
        return None, "torch.distributed batch_isend_irecv (--exchange torch)", None
    import importlib
    why, comm = None, None
    if r.backend != "nccl":
        why = "the gloo test hook puts every rank on one device, which RCCL refuses"
    else:
        rccl = importlib.import_module("savgol_amd.rccl")
        if not rccl.available():
            why = "librccl.so / lib/libsavgol_hip_rccl.so did not load"
    # the unique id travels OUTSIDE any try block: a rank-0 failure must not leave the other ranks inside broadcast_object_list
    uid = [None]
    if why is None and r.rank == 0:
        try:
            uid[0] = rccl.unique_id()
        except Exception as exc:                                  # noqa: BLE001
            why = f"ncclGetUniqueId: {exc}"
    r.dist.broadcast_object_list(uid, src=0)
    if why is None and uid[0] is None:
        why = "rank 0 could not create the unique id"
    if why is None:
        try:
            comm = rccl.Comm(r.world, r.rank, uid[0])
        except Exception as exc:                                  # noqa: BLE001
            why = f"ncclCommInitRank: {exc}"
    if not r.agree(why is None):
        if r.rank == 0:
            print(json.dumps({"error": "--exchange c: the C RCCL exchange did not come up on every rank; run with --exchange torch for the "
                                       "torch.distributed form", "rank0_reason": why}), flush=True)
        r.finish(3)
    return comm, "savgol2d_rowband_exchange_rccl (C ABI: one pack launch + ncclSend/ncclRecv per neighbour, own stream)", comm.count()


def run_image(r):
    """BASELINE config 4 as the main line: whole frames per GPU (no collective), or --rowband: one row band of every frame per GPU with the
    ny-row halos traded point to point"""
    args, sg, dev = r.args, r.sg, r.dev
    size, n = args.size, 7
    f2 = sg.Filter2D(n, n, 3)
    extra_top = {}
    if args.rowband:
        import importlib
        rowband = importlib.import_module("savgol_amd.rowband")
        Nimg = min(args.images, 128)
        band = rowband.RowBand(size * r.world, n, r.rank, r.world)          # weak scaling: a (size*world)-row frame, `size` rows per GPU
        rows = band.hi - band.lo
        local_rows = torch.empty((Nimg * rows, size), dtype=torch.float32, device=dev)
        sg.synth(local_rows, channel0=band.lo)
        local_band = local_rows.view(Nimg, rows, size)
        comm, exchange, ranks = None, "none (one rank)", None
        if r.world > 1:
            comm, exchange, ranks = rowband_comm(r, args.exchange)
        extra_top = {"exchange": exchange, "rccl_ranks": ranks}

        def apply_fn(frames):
            k, rr, c = frames.shape
            o = torch.empty_like(frames)
            f2.apply_batch(frames, o, rr, c, k, boundary=1, method=args.method)
            return o

        def step(events):
            e0, e1 = ev(), ev()
            # through the C ABI (savgol2d_apply_batch_f32 on the band while the halos travel, then savgol2d_apply_rowband_edges_f32);
            # bands thinner than 2 n rows (tiny test shapes) take the Python form
            e0.record()
            if band.thin:
                band.apply_overlapped(local_band, apply_fn)
            else:
                band.apply_c(f2, local_band, boundary=1, method=args.method, comm=comm)
            e1.record()
            if events is not None:
                events.append((e0, e1))
        el, events = r.timed_region(step)
        pix_rank = Nimg * rows * size
        cfg = {"workload": f"BASELINE config 4 shape, row-band split: {Nimg} frames of {size * r.world} x {size} fp32, one {rows}-row band per GPU, "
                           f"n=7, order 3, CONSTANT; per step: {n}-row halos to both neighbours (RCCL point to point), band filtered meanwhile, edge strips redone",
               "sharding": "row bands, nearest-neighbour halo exchange", "exchange": exchange}
        ceil = {}
    else:
        Nimg = args.images
        x = torch.empty((Nimg * size, size), dtype=torch.float32, device=dev); sg.synth(x, channel0=r.rank * Nimg * size)
        y = torch.empty_like(x)
        ceil = copy_ceiling(sg, x, y)

        def step(events):
            for b in (0, 1, 2):
                e0, e1 = ev(), ev()
                e0.record(); f2.apply_batch(x, y, size, size, Nimg, boundary=b, method=args.method); e1.record()
                if events is not None:
                    events.append((e0, e1))
        el, events = r.timed_region(step)
        pix_rank = 3 * Nimg * size * size
        cfg = {"workload": f"BASELINE config 4: {Nimg} images x {size}x{size} fp32 per GPU, n=7, order 3, one pass per boundary mode "
                           f"(VALID, CONSTANT, REFLECT) per step, method {args.method}", "sharding": "images, no collective"}
    if r.rank == 0:
        lms = [p.elapsed_time(q) for p, q in events]
        ms = float(np.mean(lms))
        per_launch_pix = pix_rank if args.rowband else pix_rank // 3
        out = {"metric": "Mpix/s filtered (2-D, hw=7, order 3)", "value": round(pix_rank * args.steps * r.world / el / 1e6, 1), "unit": "Mpix/s", **r.common,
               "ms_per_step": round(el / args.steps * 1e3, 4), "dtype": "f32", "config": cfg, **extra_top,
               "roofline": add_ceiling(roofline(8.0 * per_launch_pix, ms, lms, kernel=IMAGE_KERNEL if args.method == 2 else "sg2d_dense_roll_kernel<7>"), ceil)}
        if args.method == 2 and not args.rowband:
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = pmc_traffic(8.0 * per_launch_pix, "r*_2d_config4_pmc_summary.json", SOURCES_2D)
        pr = r.per_rank_fields(8.0 * per_launch_pix)
        if pr:
            if args.rowband:
                pr.pop("rccl_ranks", None)                  # the row-band line reports the exchange's own communicator (extra_top)
            out.update(pr)
        if r.world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_reference("image")
        emit(out)
    r.finish()


def run_config5(r):
    """BASELINE config 5 as the main line: every GPU its 4096-channel fp64 slice, 1024-channel chunks per launch"""
    args, sg, dev = r.args, r.sg, r.dev
    channels, chunk, length = args.c5_channels, args.c5_chunk, 1 << 22
    x, y, resident, full_out = config5_buffers(sg, channels, chunk, length, r.rank, dev)
    f = sg.Filter(N, M, 2, 1.0, 0)
    ceil = copy_ceiling(sg, x[:chunk], y[:chunk], reps=3)
    # the tolerance config 5 states, in the call (savgol_apply_batch_f64_tol); --f64-exact: the default tap-by-tap path (1e-12) instead
    exact = args.f64_exact
    call_kw = {"flags": 0} if exact else {"rel_tol": C5_TOL}

    def step(events):
        for c0 in range(0, resident, chunk):
            e0, e1 = ev(), ev()
            e0.record()
            f.apply_batch(x[c0:c0 + chunk], y[c0:c0 + chunk] if full_out else y, chunk, length, dtype="f64", **call_kw)
            e1.record()
            if events is not None:
                events.append((e0, e1))
    el, events = r.timed_region(step)
    if r.rank == 0:
        lms = [p.elapsed_time(q) for p, q in events]
        ms = float(np.mean(lms))
        out = {"metric": "Msamples/s filtered (1D batch fp64, hw=32, poly=4, derivative=2) + % HBM roofline",
               "value": round(resident * length * args.steps * r.world / el / 1e6, 1), "unit": "Msamples/s", **r.common,
               "ms_per_step": round(el / args.steps * 1e3, 4), "dtype": "f64",
               "config": {"workload": f"BASELINE config 5: {resident} channels x {length} fp64 samples per GPU ({channels} = 32768/8 asked for; "
                                      f"{resident * length * 8 / 1e9:.1f} GB resident input), n=32, m=4, d=2, POLYNOMIAL, {chunk}-channel chunks per launch, "
                                      + ("output slice resident" if full_out else "one chunk-sized output buffer reused"),
                          "channels_per_gpu": resident, "length": length, "sharding": "channels, no collective"},
               "roofline": add_ceiling(with_traffic(roofline(16.0 * chunk * length, ms, lms, kernel="sg1d_center_kernel<double,32>" if exact else "sg1d_center_moment64_kernel<32,5>"),
                                                    "r*_1d_f64_n32_pmc_summary.json" if exact else "r*_1d_f64m_n32_pmc_summary.json"), ceil)}
        out.update(r.per_rank_fields(16.0 * chunk * length))
        out["config"]["call"] = "savgol_apply_batch_f64_ex(flags=0): 1e-12 of the fp64 oracle" if exact else f"savgol_apply_batch_f64_tol(rel_tol={C5_TOL:g}): the bar the config states"
        if r.world == 1 and not args.no_cpu:
            from oracle import sgo
            sample = [0, chunk - 1]
            c0 = resident - chunk
            got = (y[c0:c0 + chunk] if full_out else y)[sample].cpu().numpy()
            ref = sgo.Filter(N, M, 2, 1.0, 0).apply_f64(x[c0:c0 + chunk][sample].cpu().numpy())
            err = normwise(got, ref)
            assert err < (1e-12 if exact else C5_TOL), f"parity lost: normwise error {err}"
            out["parity_normwise_vs_fp64_oracle"] = err
            out["cpu_baseline"] = cpu_baseline(1 << 20, N, M, 2, budget_s=8.0, all_cores=False)
            out["cpu_baseline"]["sample"] += " (fp32: the reference has no fp64 path)"
        emit(out)
    r.finish()


def placement_spread(sg, x, filters, ch, length, alg_bytes):
    """the headline's passes on up to three fresh buffer pairs (a function of its own: every temporary dies on return -- config 5's slice needs the memory)"""
    try:
        free_b, _ = torch.cuda.mem_get_info()
        pairs = int(min(3, (free_b - (8 << 30)) // (2 * x.numel() * 4))) if free_b > (8 << 30) else 0
        fr, keep = [], []
        for _ in range(pairs):
            x2 = torch.empty_like(x); y2 = torch.empty_like(x)
            keep += [x2, y2]                                             # alive until the end: the next pair must land on other physical pages
            sg.synth(x2, channel0=0)
            for _w in range(2):                                          # the first launches over a fresh pair are slower whatever they are
                filters[0].apply_batch(x2, y2, ch, length)
            ms2 = timed(lambda: [flt.apply_batch(x2, y2, ch, length) for flt in filters], reps=3, warm=1) / len(filters)
            fr.append(round(alg_bytes / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
        if not fr:
            return {"fresh_pairs": 0, "note": "not enough free memory for another pair of buffers"}
        return {"fresh_pairs": len(fr), "frac_of_each": fr, "frac_min": min(fr), "frac_max": max(fr),
                "note": "the same four passes on fresh allocations of the two buffers inside this process, after the timed region: one run's frac is one "
                        "placement (tools/placement_1d.py, profiles/r05_placement_1d.txt)"}
    except Exception as e:                                               # noqa: BLE001 -- diagnostic only
        return {"error": f"{type(e).__name__}: {e}"}


def run_headline(r):
    """BASELINE config 2, the headline: 4096 channels x 2^20 fp32 per GPU, n=32, m=4, one launch per boundary mode per step"""
    args, sg, dev = r.args, r.sg, r.dev
    ch, length = args.channels, args.length
    ch0 = r.rank * ch
    if args.total_channels:
        # a FIXED channel count split over the ranks (strong scaling; uneven when it does not divide: savgol_hip_shard_range gives the first
        # total % world ranks one channel more) -- the plumbing check for ragged shards (tools/check_multirank_plumbing.sh)
        ch0, hi = sg.shard_range(args.total_channels, r.world, r.rank)
        ch = hi - ch0
    modes = [0, 1, 2, 3]
    x = torch.empty((ch, length), dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    sg.synth(x, channel0=ch0)                               # generated in HBM, never crosses PCIe
    filters = [sg.Filter(N, M, D, 1.0, mode) for mode in modes]
    torch.cuda.synchronize()
    ceil = copy_ceiling(sg, x, y)                           # the same two buffers, the same process, before the timed region

    def step(events):
        for f in filters:
            e0, e1 = ev(), ev()
            e0.record(); f.apply_batch(x, y, ch, length); e1.record()
            if events is not None:
                events.append((e0, e1))
    elapsed, events = r.timed_region(step)

    if r.rank == 0:
        launches_ms = [a.elapsed_time(b) for a, b in events]
        avg_ms = float(np.mean(launches_ms))
        alg_bytes = 8.0 * ch * length                        # 4 B read + 4 B written per sample
        samples = float(len(modes)) * (args.total_channels if args.total_channels else ch * r.world) * length * args.steps
        import ctypes as C
        tab = (C.c_float * 400)()
        terms = sg.lib().savgol_hip_momenth_table(filters[0].ptr, tab)
        kernel = f"sg1d_center_momenth_kernel<{N},{terms}>" if terms > 0 else f"sg1d_center_kernel<float,{N}>"
        traffic, traffic_src = pmc_traffic(alg_bytes)
        out = {
            "metric": "Msamples/s filtered (1D batch, hw=32, poly=4) + % HBM roofline",
            "value": round(samples / elapsed / 1e6, 1), "unit": "Msamples/s", **r.common,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "dtype": "f32",
            "config": {"workload": f"BASELINE config 2: {ch} channels x {length} fp32 samples per GPU, half_window={N}, "
                                   f"poly_order={M}, derivative={D}, one pass per boundary mode "
                                   "(POLYNOMIAL, REFLECT, PERIODIC, CONSTANT) per step",
                       "channels_per_gpu": ch, "length": length, "sharding": "channels, no collective",
                       **({"total_channels": args.total_channels, "split": "savgol_hip_shard_range (rank 0's share shown in channels_per_gpu)"} if args.total_channels else {})},
            "roofline": add_ceiling({**roofline(alg_bytes, avg_ms, launches_ms, kernel=kernel), "traffic": traffic, "traffic_source": traffic_src}, ceil),
            # where the two 16 GiB buffers landed: the same launch runs 5.2-5.75 ms depending on their physical placement
            # (DESIGN 4.1 "Run-to-run spread"); with the addresses a 0.75 run and a 0.83 run can be told apart
            "buffers": {"x": hex(x.data_ptr()), "y": hex(y.data_ptr()), "bytes_each": x.numel() * 4},
        }
        out.update(r.per_rank_fields(alg_bytes))
        if args.total_channels:
            out["scaling"] = "strong"
        if r.world == 1 and not args.no_cpu:
            # CPU leg (rank 0, N=1 only): the reference timed on this host + parity spot checks of what was just timed
            # (last mode run = CONSTANT) against the CPU oracle and against the reference's own fp32 output -- the only place bench.py touches oracle/
            out["cpu_baseline"] = cpu_baseline(length)
            from oracle import sgo
            sample = [0, ch // 2, ch - 1]
            xs = x[sample].cpu().numpy()
            ref64 = sgo.Filter(N, M, D, 1.0, modes[-1]).apply_f64(xs.astype(np.float64))
            ref32, which = reference_fp32_1d(xs, N, M, D, 1.0, modes[-1])
            out.update(parity_fields(y[sample].cpu().numpy(), ref64, ref32, which))
            assert out["parity_normwise_vs_fp64_oracle"] < 1e-6, f"parity lost: normwise error {out['parity_normwise_vs_fp64_oracle']}"
            assert out["parity_normwise_vs_reference_fp32"] < 1e-6, f"parity lost vs the reference's own output: {out['parity_normwise_vs_reference_fp32']}"
            # a derivative filter through the same kernel: 1e-6, or 1.1 x the reference's own error on these channels where that is larger
            fd = sg.Filter(N, M, 1, 1.0, 0)
            fd.apply_batch(x, y, ch, length); torch.cuda.synchronize()
            refd = sgo.Filter(N, M, 1, 1.0, 0).apply_f64(xs.astype(np.float64))
            refd32, _ = reference_fp32_1d(xs, N, M, 1, 1.0, 0)
            d1 = parity_fields(y[sample].cpu().numpy(), refd, refd32, which)
            out["parity_d1"] = {k: v for k, v in d1.items() if k != "reference_fp32_from"}
            assert d1["parity_normwise_vs_fp64_oracle"] < max(1e-6, 1.1 * d1["reference_fp32_own_error_vs_fp64_oracle"])
        if r.world == 1:
            # secondary figures, outside the timed region
            L = sg.lib()
            sec = {}
            for name, opt in (("plain_65_tap_sum", sg.SAVGOL_HIP_OPT_PLAIN_SUMMATION), ("bit_identical_mode", sg.SAVGOL_HIP_OPT_REFERENCE_SUMMATION)):
                if L.savgol_hip_set_option(opt, 1) != 0:
                    continue
                try:
                    ms = timed(lambda: [flt.apply_batch(x, y, ch, length) for flt in filters], reps=3, warm=1) / len(filters)
                    sec[name] = {"Msamples_per_s": round(ch * length / ms / 1e3, 1), "avg_launch_ms": round(ms, 4),
                                 "roofline_frac": round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                    if name == "bit_identical_mode" and not args.no_cpu:
                        sample = [0, ch - 1]
                        ref32, _ = reference_fp32_1d(x[sample].cpu().numpy(), N, M, D, 1.0, modes[-1])
                        assert np.array_equal(y[sample].cpu().numpy().view(np.uint32), ref32.view(np.uint32)), "bit-identical mode lost"
                finally:
                    L.savgol_hip_set_option(opt, 0)
            if "plain_65_tap_sum" in sec:
                sec["plain_65_tap_sum"]["note"] = "SAVGOL_HIP_OPT_PLAIN_SUMMATION: sg1d_center_kernel<float,32>, all 65 taps one by one (the default replaces 32 of them by block moments)"
            if "bit_identical_mode" in sec:
                sec["bit_identical_mode"]["note"] = ("reference summation order (four chains, separate multiply and add): outputs equal the reference "
                                                     "library's bit for bit")
            out.update(sec)
        if r.world == 1 and not args.no_extra:
            # PLACEMENT SPREAD (round 5, profiles/EXPERIMENTS.md R5.9): the same launch on a few FRESH buffer pairs inside this process, the timed pair
            # kept alive so that new physical pages back the new ones -- how much of this run's `frac` is where its two 16 GiB buffers happen to
            # sit.  Outside the timed region; `roofline.frac` above is the timed pair's and nothing else.
            out["roofline"]["placement_spread"] = placement_spread(sg, x, filters, ch, length, alg_bytes)
            del x, y
            torch.cuda.empty_cache()
            extra = {"build": build_facts(sg)}
            for name, fn in (("config1", lambda: bench_config1(sg, args.no_cpu)), ("config3", lambda: bench_stream(sg, args)),
                             ("config4", lambda: bench_image(sg, args)), ("config4_rowband", lambda: bench_rowband_ring_of_one(sg, args)),
                             ("config5_slice", lambda: bench_config5_slice(sg, args))):
                try:
                    extra[name] = fn()
                except Exception as e:                               # an extra must never take the headline down
                    extra[name] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
            out["extra"] = extra
        emit(out)
    r.finish()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--channels", type=int, default=4096, help="channels per GPU")
    ap.add_argument("--length", type=int, default=1 << 20)
    ap.add_argument("--total-channels", type=int, default=0, help="batch1d: a fixed channel count split over the ranks by savgol_hip_shard_range (uneven shards when it "
                                                                   "does not divide; scaling 'strong') instead of --channels per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs and the oracle spot checks")
    ap.add_argument("--no-extra", action="store_true", help="headline only: skip the configs 1/3/4/5 'extra' section")
    ap.add_argument("--workload", choices=["batch1d", "batch1d_f64", "stream", "image"], default="batch1d",
                    help="batch1d = the headline (BASELINE config 2); batch1d_f64 = config 5's per-GPU slice (4096 channels x 2^22 fp64, "
                         "n=32, d=2, POLYNOMIAL, 1024-channel chunks); stream / image = configs 3 / 4")
    ap.add_argument("--streams", type=int, default=65536)
    ap.add_argument("--ticks", type=int, default=4096)
    ap.add_argument("--images", type=int, default=512, help="2-D: frames per pass (BASELINE config 4 has 512 = 34 GB in + 34 GB out)")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--method", type=int, default=2, help="2-D: 1 = dense (bit-exact), 2 = separable")
    ap.add_argument("--exchange", choices=["c", "torch"], default="c", help="--rowband: who moves the halos -- c = savgol2d_rowband_exchange_rccl "
                    "(lib/libsavgol_hip_rccl.so, the C entry point INTEGRATION.md documents; the job EXITS NON-ZERO when it cannot come up on every rank), "
                    "torch = torch.distributed batch_isend_irecv (the explicit fallback)")
    ap.add_argument("--rowband", action="store_true", help="--workload image: split every frame into one row band per GPU and trade the "
                                                           "ny-row halos with the neighbours (RCCL point to point) instead of sharding whole frames")
    ap.add_argument("--c5-channels", type=int, default=4096, help="config 5: channels per GPU (32768 / 8)")
    ap.add_argument("--c5-chunk", type=int, default=1024)
    ap.add_argument("--f64-exact", action="store_true", help="--workload batch1d_f64: the default tap-by-tap fp64 path (1e-12 of the oracle) instead of "
                                                            "savgol_apply_batch_f64_tol(rel_tol=1e-6), the tolerance config 5 states")
    args = ap.parse_args()

    # ---- N ranks: start them from here, BEFORE anything in this process touches the GPU (never re-exec after that) ----
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    r = Run(args)
    if r.dry:
        return dry_run(r, args)
    r.attach_gpu()
    {"stream": run_stream, "image": run_image, "batch1d_f64": run_config5, "batch1d": run_headline}[args.workload](r)


if __name__ == "__main__":
    main()
