import numpy as np


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def normwise(a, b):
    """max|a-b| / max|b| -- the parity metric of SURVEY.md section 7 (hard part 3)"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


def same_bits(a, b):
    """fp32 arrays equal bit for bit (so -0.0 != +0.0 and NaN payloads count), shapes included"""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def fuzz(seed, iters):
    """(seed, iterations) of a randomized test: the committed defaults, or a soak run's override --
    SAVGOL_FUZZ_SEED shifts every seed, SAVGOL_FUZZ_SCALE multiplies every iteration count (tools/soak_gpu.sh)."""
    import os
    return seed + int(os.environ.get("SAVGOL_FUZZ_SEED", "0")), int(iters * float(os.environ.get("SAVGOL_FUZZ_SCALE", "1")))


# ---- fp32 parity bars (round 5: every bar is north_star's 1e-6 unless the REFERENCE ITSELF is further than that from the exact answer) ----
BAR = 1e-6           # north_star: "within 1e-6 relative"
REF_SLACK = 1.1      # where the reference's own fp32 output is already > 1e-6 from the double answer: 1.1 x its error, the case recorded


def fp32_bar(ref_err=0.0):
    """bar for an fp32 kernel against the double oracle: 1e-6, or 1.1 x the reference's own fp32 error (reference order emulated by the
    oracle, or the golden fixture) where the reference itself exceeds 1e-6 -- tools/parity_margins.py lists the cases where that happens"""
    return max(BAR, REF_SLACK * float(ref_err))


def check(value, bar, label):
    """assert value < bar; with SAVGOL_PARITY_LOG=path every comparison is appended there (tools/parity_margins.py prints the worst per test)"""
    import json
    import os
    path = os.environ.get("SAVGOL_PARITY_LOG")
    if path:
        with open(path, "a") as fh:
            fh.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], "label": str(label), "value": float(value), "bar": float(bar)}) + "\n")
    if os.environ.get("SAVGOL_PARITY_NOASSERT") != "1":     # a margin survey logs every comparison, failing ones included
        assert value < bar, (label, value, bar)
