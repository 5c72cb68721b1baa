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
    return json.loads(lines[0])


def test_single_gpu_line():
    d = run([sys.executable, "bench.py", "--channels", "256", "--length", "262144", "--steps", "3", "--warmup", "1"])
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["unit"] == "Msamples/s" and d["dtype"] == "f32" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and d["roofline"]["bound"] == "hbm"
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["cores"] == 1
    assert d["parity_normwise_vs_fp64_oracle"] < 1e-6
    assert abs(d["value"] - 4 * 256 * 262144 * 3 / (d["ms_per_step"] * 3 * 1e-3) / 1e6) / d["value"] < 0.01


def test_two_ranks_weak_scaling_plumbing():
    port = "29617"
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", port, "bench.py", "--gpus", "2", "--channels", "256", "--length", "262144", "--steps", "3",
             "--warmup", "1"], env={"SAVGOL_BENCH_BACKEND": "gloo", "SAVGOL_BENCH_DEVICE": "0"})
    assert REQUIRED <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert "cpu_baseline" not in d                                  # CPU leg only at N=1
    # whole-job aggregate over both ranks: 2 x (4 modes x channels x length x steps) samples in the max-over-ranks time
    assert abs(d["value"] - 2 * 4 * 256 * 262144 * 3 / (d["ms_per_step"] * 3 * 1e-3) / 1e6) / d["value"] < 0.01
