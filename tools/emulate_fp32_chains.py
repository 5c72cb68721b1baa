"""How should an fp32 sliding dot product be split into partial sums?  Emulates the device kernel's arithmetic on the CPU (a fused
multiply-add = the exact double product + one rounding to fp32) for several ways of dealing the 2n+1 taps to chains, on the same
synthetic signals tests/test_gpu_1d.py::test_fp32_kernels_against_the_reference_s_own_fp32_error uses, and prints for each half
window the WORST error over poly_order <= 6, derivative <= 2 as a multiple of the reference's own fp32 error (four round-robin chains,
products and adds rounded separately: src/savgolFilter.c:547-580), both normwise against the double-accumulation sum.
    python tools/emulate_fp32_chains.py > profiles/r04_fp32_chains.txt"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sgo

f32 = np.float32


def normwise(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b)) / np.max(np.abs(b)))


def fma(x, w, c):
    return (x.astype(np.float64) * np.float64(w) + c.astype(np.float64)).astype(f32)


def chains(x, w, groups, fused=True):
    """groups: lists of tap indices, each summed in order as its own chain; the chains are joined pairwise, in order"""
    L = x.shape[1] - len(w) + 1
    parts = []
    for g in groups:
        acc = None
        for k in g:
            xs = x[:, k:k + L]
            if acc is None:
                acc = (xs * f32(w[k])).astype(f32)
            elif fused:
                acc = fma(xs, w[k], acc)
            else:
                acc = (acc + (xs * f32(w[k])).astype(f32)).astype(f32)
        parts.append(acc)
    while len(parts) > 1:
        parts = [(parts[i] + parts[i + 1]).astype(f32) if i + 1 < len(parts) else parts[i] for i in range(0, len(parts), 2)]
    return parts[0]


def reference_groups(ws):
    r = ws & 3
    g = [[], [], [], []]
    for k in range(r):
        g[k].append(k)
    for k in range(r, ws):
        g[(k - r) % 4].append(k)
    return g


def main():
    names = ["one chain", "two halves", "even/odd", "three thirds", "three round robin", "four quarters", "four round robin"]
    print("worst error over poly_order <= 6, derivative <= 2 as a multiple of the reference's own fp32 error (floor 2.5e-7), interior outputs")
    print("  n  " + "  ".join(f"{s:>17s}" for s in names))
    for n in (5, 8, 12, 16, 20, 23, 24, 25, 28, 31, 32):
        x = sgo.synth_f32(3 * n, 5, 40000 + 17 * n)
        ws = 2 * n + 1
        variants = {
            "one chain": [list(range(ws))],
            "two halves": [list(range(n + 1)), list(range(n + 1, ws))],
            "even/odd": [list(range(0, ws, 2)), list(range(1, ws, 2))],
            "three thirds": [list(range(i * ws // 3, (i + 1) * ws // 3)) for i in range(3)],
            "three round robin": [list(range(i, ws, 3)) for i in range(3)],
            "four quarters": [list(range(i * ws // 4, (i + 1) * ws // 4)) for i in range(4)],
            "four round robin": [list(range(i, ws, 4)) for i in range(4)],
        }
        worst = dict.fromkeys(names, 0.0)
        for m in range(0, 7):
            for d in range(0, min(m, 2) + 1):
                w = np.asarray(sgo.weights(n, m, d)[0] if isinstance(sgo.weights(n, m, d), tuple) else sgo.weights(n, m, d), dtype=f32).ravel()[:ws]
                ref64 = sum(x[:, k:k + x.shape[1] - ws + 1].astype(np.float64) * np.float64(w[k]) for k in range(ws))
                e_ref = normwise(chains(x, w, reference_groups(ws), fused=False), ref64)
                for name in names:
                    worst[name] = max(worst[name], normwise(chains(x, w, variants[name]), ref64) / max(e_ref, 2.5e-7))
        print(f"{n:3d}  " + "  ".join(f"{worst[s]:17.2f}" for s in names), flush=True)


if __name__ == "__main__":
    main()
