"""`python bench.py --gpus N` must start its N ranks itself (the flag was parsed and ignored, so the driver's
multi-GPU run did one GPU's work and printed n_gpus: 1).  Checked here on CPU: SAVGOL_BENCH_DRYRUN=1 keeps the launch, the
gloo rendezvous on 127.0.0.1, the barrier and the max-over-ranks reduce, and skips the GPU work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)


def test_gpus_flag_launches_that_many_ranks():
    r = _run(["--gpus", "2"], {"SAVGOL_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["max_over_ranks"] == 2.0          # rank 1 contributed 1.0 + rank


def test_mismatch_between_flag_and_world_size_is_an_error():
    r = _run(["--gpus", "4"], {"SAVGOL_BENCH_DRYRUN": "1", "WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_rowband_two_ranks_dry_run_plans_the_bands():
    """--workload image --rowband --gpus 2: the self-launch, the band plan (the two bands tile the doubled frame, rank 0
    has a neighbour below and none above) and the choice of exchange travel through the dry run; nothing re-execs after GPU init because
    nothing touches the GPU at all here."""
    r = _run(["--gpus", "2", "--workload", "image", "--rowband", "--size", "256"], {"SAVGOL_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rowband"] is True and d["exchange"] == "c"
    assert d["frame_rows"] == 512 and d["rows_over_ranks"] == 512 and d["rank0_band"] == [0, 256]
    assert d["rank0_neighbours"] == {"up": False, "down": True}
