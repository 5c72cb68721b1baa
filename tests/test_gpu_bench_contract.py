"""bench.py's contract with the driver: one JSON line with the required keys at N=1, and the N>1 launch
(`python -m torch.distributed.run ... bench.py --gpus N`) -- exercised here with 2 ranks sharing the one GPU of the
test box (gloo for the 8-byte timing reduce), small shapes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"}


def run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    d["_stdout_tail_4k"] = r.stdout[-4096:]
    return d


def test_single_gpu_line():
    d = run([sys.executable, "bench.py", "--channels", "256", "--length", "262144", "--steps", "3", "--warmup", "1"])
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["unit"] == "Msamples/s" and d["dtype"] == "f32" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and d["roofline"]["bound"] == "hbm"
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["cores"] == 1
    assert d["parity_normwise_vs_fp64_oracle"] < 1e-6
    assert abs(d["value"] - 4 * 256 * 262144 * 3 / (d["ms_per_step"] * 3 * 1e-3) / 1e6) / d["value"] < 0.01
    # round 5: the distance from the REFERENCE's own fp32 output, and the copy ceiling measured in this process
    assert d["parity_normwise_vs_reference_fp32"] < 1e-6 and d["reference_fp32_own_error_vs_fp64_oracle"] < 1e-6 and "reference_fp32_from" in d
    roof = d["roofline"]
    assert {"copy_frac", "frac_of_copy", "read_ceiling_frac", "kernel_read_only_frac"} <= set(roof)
    assert 0.3 < roof["copy_frac"] < 1.0 and 0.3 < roof["read_ceiling_frac"] < 1.0 and abs(roof["frac_of_copy"] - roof["frac"] / roof["copy_frac"]) < 1e-3
    assert abs(roof["kernel_read_only_frac"] - 0.5 * roof["frac"]) < 1e-3          # N x sizeof(T) / t / 8e12, the kernel's own figure
    # round 6: a compact per-config summary is the LAST key of the line, inside the tail the driver keeps, and no prose rides along
    tail = d["_stdout_tail_4k"]
    assert '"summary"' in tail and list(d)[-2] == "summary", list(d)[-3:]
    sm = d["summary"]
    assert len(json.dumps(sm)) <= 1400, len(json.dumps(sm))
    for key in ("c1", "c2", "c3_fused", "c3_bit_exact", "c4_VALID", "c4_CONSTANT", "c4_REFLECT", "c4_rowband", "c5", "c5_in_place"):
        if key == "c4_rowband" and "skipped" in d["extra"]["config4_rowband"]:
            continue
        assert key in sm and set(sm[key]) >= {"frac", "ms", "parity"}, (key, sm.get(key))
        assert sm[key]["frac"] is not None and sm[key]["ms"] is not None, (key, sm[key])
    assert "errors" not in sm, sm
    assert sm["c3_bit_exact"]["parity"] == 0.0 and sm["c2"]["parity"] < 1e-6 and sm["c5"]["parity"] < 1e-6
    assert '"note"' not in json.dumps({k: v for k, v in d.items() if k != "_stdout_tail_4k"})
    # round 5: the same passes on fresh buffer pairs inside the process (one run's frac is one placement of its two buffers)
    sp = roof["placement_spread"]
    assert "error" not in sp and sp["fresh_pairs"] >= 1 and all(0.05 < v < 1.0 for v in sp["frac_of_each"]), sp
    ex = d["extra"]
    assert ex["build"]["library"].endswith("libsavgol_hip.so") and "library_sha256_16" in ex["build"]
    for leg, keys in (("config1", ()), ("config3", ("block_push",)), ("config4", ("modes", "CONSTANT"))):
        node = ex[leg]
        assert "error" not in node, node
        for k in keys:
            node = node[k]
        assert node["parity_normwise_vs_reference_fp32"] < 2e-6 and node["parity_normwise_vs_fp64_oracle"] < 2e-6, (leg, node)
    assert ex["config3"]["block_push"]["roofline"]["copy_frac"] > 0.3 and ex["config4"]["modes"]["CONSTANT"]["roofline"]["frac_of_copy"] > 0.3
    rb = ex["config4_rowband"]
    assert "error" not in rb, rb
    if "skipped" not in rb:
        assert rb["rccl_ranks"] == 1 and rb["exchange_ms"] > 0 and rb["band_ms"] > 0 and rb["edge_strips_ms"] > 0 and rb["step_ms"] > 0
        assert rb["parity_normwise_vs_fp64_oracle"] < 1e-6


def test_exchange_c_fails_loudly_when_it_cannot_come_up():
    """--rowband --exchange c must END THE JOB (non-zero exit, a JSON error line) when the C RCCL exchange cannot come up on every rank -- here
    because both ranks sit on one GPU, which RCCL refuses (the gloo hook) -- instead of silently timing torch.distributed's P2P (round 4);
    --exchange torch on the same ranks is the explicit fallback and runs."""
    def cmd(port, exchange):
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
                "bench.py", "--gpus", "2", "--workload", "image", "--rowband", "--images", "2", "--size", "512", "--steps", "2", "--warmup", "1", "--exchange", exchange]
    e = dict(os.environ); e.update({"SAVGOL_BENCH_BACKEND": "gloo", "SAVGOL_BENCH_DEVICE": "0"})
    r = subprocess.run(cmd("29619", "c"), cwd=ROOT, capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode != 0
    assert "--exchange c" in r.stdout and "--exchange torch" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    d = run(cmd("29621", "torch"), env={"SAVGOL_BENCH_BACKEND": "gloo", "SAVGOL_BENCH_DEVICE": "0"})
    assert d["n_gpus"] == 2 and "torch.distributed" in d["exchange"] and d["rccl_ranks"] is None
