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
