#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes of one kernel into profiles/<name>.json.

    python tools/pmc_summary.py --kernel 'sg1d_center_kernel<float, 32>' --alg-bytes 34359738368 \
        --fetch gpurun_out/r2_pmc_fetch --write gpurun_out/r2_pmc_write --sq gpurun_out/r2_pmc_sq \
        --command '...' --workload '...' --out profiles/r02_1d_f32_n32_pmc_summary.json

Each --fetch/--write/--sq directory is the -d output of ONE `rocprofv3 --kernel-trace --pmc <set> -- <cmd>` run (the
counter sets do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots).  HBM bytes per launch are
FETCH_SIZE[KiB]*1024*2 (gfx950 tallies the 128-B requests of a wide streaming read at 64 B) + WRITE_SIZE[KiB]*1024, as that
guide's HBM section prescribes.  The summary records the sha256 of the kernel sources it was taken on
(`kernel_source_sha`, same function bench.py uses), so a stale summary is never reported as the current traffic."""
import argparse
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha(files=None):
    """sha256 over the sources that define the 1-D batch kernels (bench.py compares this with the summary's)."""
    files = files or ["sg_k1d.hpp", "sg_k1d_host.hpp", "sg_k1d_inst.hip", "sg_k1d_momenth.hpp", "sg_k1d_momenth.hip", "sg_k1d_moment64.hpp", "sg_k1d_moment64.hip",
                      "sg_k1d_moment_fit.cpp", "sg_pk.hpp", "sg_api_1d.cpp"]
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "savitzky-golay-filter_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def read_pass(d, kernel):
    """-> ({counter: mean value per dispatch}, mean duration ms, dispatches)"""
    vals, dur = defaultdict(list), {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if kernel not in row["Kernel_Name"]:
                continue
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
    n = len(dur)
    return {k: sum(v) / len(v) for k, v in vals.items()}, (sum(dur.values()) / n if n else None), n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--alg-bytes", type=float, required=True)
    ap.add_argument("--fetch"); ap.add_argument("--write"); ap.add_argument("--sq")
    ap.add_argument("--command", default=""); ap.add_argument("--workload", default="")
    ap.add_argument("--sources", nargs="*", help="kernel source files under csrc/ to hash (default: the 1-D batch kernel's)")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    out = {"command": a.command, "kernel": a.kernel, "workload": a.workload, "algorithmic_bytes_per_launch": a.alg_bytes,
           "kernel_source_sha": kernel_source_sha(a.sources), "kernel_source_files": a.sources or "1-D batch default",
           "avg_duration_ms_under_pmc": {}}
    traffic = 0.0
    if a.fetch:
        v, ms, n = read_pass(a.fetch, a.kernel)
        out["FETCH_SIZE_KiB_raw"] = v["FETCH_SIZE"]; out["fetch_bytes_corrected_x2"] = v["FETCH_SIZE"] * 1024 * 2
        out["avg_duration_ms_under_pmc"]["fetch_pass"] = ms; out["dispatches_fetch_pass"] = n
        traffic += out["fetch_bytes_corrected_x2"]
    if a.write:
        v, ms, n = read_pass(a.write, a.kernel)
        out["WRITE_SIZE_KiB_raw"] = v["WRITE_SIZE"]; out["write_bytes"] = v["WRITE_SIZE"] * 1024
        out["avg_duration_ms_under_pmc"]["write_pass"] = ms
        traffic += out["write_bytes"]
    if a.fetch and a.write:
        out["hbm_traffic_bytes_per_launch"] = traffic
        out["traffic_over_algorithmic"] = traffic / a.alg_bytes
    if a.sq:
        v, ms, n = read_pass(a.sq, a.kernel)
        out["avg_duration_ms_under_pmc"]["sq_pass"] = ms
        out["sq_counters_avg_per_launch"] = v
        d = {}
        if "GRBM_GUI_ACTIVE" in v and ms:
            d["effective_clock_GHz"] = v["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3) / 1e9
        if "SQ_INSTS_VALU" in v:
            d["valu_instr_per_simd"] = v["SQ_INSTS_VALU"] / 1024
            if "effective_clock_GHz" in d:
                d["cycles_per_valu_instr_per_simd"] = d["effective_clock_GHz"] * 1e9 * ms * 1e-3 / d["valu_instr_per_simd"]
        if "SQ_WAVE_CYCLES" in v:
            for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
                if k in v:
                    d[k.lower()[3:] + "_over_wave_cycles"] = v[k] / v["SQ_WAVE_CYCLES"]
        if "SQ_LDS_BANK_CONFLICT" in v and v.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_over_lds_active"] = v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]
        out["derived"] = d
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    sys.exit(main())
