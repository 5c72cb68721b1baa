"""Config 3's block push with the output buffer at a chosen byte offset behind the input inside ONE allocation: is the placement effect of R5.9 the RELATIVE
offset of the two streams (same index -> same channel / bank when the buffers are a power of two apart)?
   python tools/placement_stream_offset.py          prints ms per push (eight in a row, median of 5) for both banks at each pad"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sg = load_package()
S, T, n = 65536, 4096, 16
nbytes = S * T * 4
pads = [0, 4096, 65536, 1 << 20, (1 << 20) + 65536, 17 << 20, 96 << 20, 256 << 20]
ev = lambda: torch.cuda.Event(enable_timing=True)
def timed8(fn):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = ev(), ev(); e0.record()
        for _ in range(8): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 8)
    return float(np.median(ts))
big = torch.empty(2 * nbytes + max(pads) + 4096, dtype=torch.uint8, device="cuda")
x = big[:nbytes].view(torch.float32).view(T, S)
sg.synth(x)
fused, exact = sg.StreamBank(S, n, 2, 1, 1e-3, fma=True), sg.StreamBank(S, n, 2, 1, 1e-3)
for rep in range(2):
    for pad in pads:
        out = big[nbytes + pad:2 * nbytes + pad].view(torch.float32).view(T, S)
        a = timed8(lambda: fused.push_block(x, T, out)); b = timed8(lambda: exact.push_block(x, T, out))
        f = lambda ms: 8.0 * S * T / (ms * 1e-3) / 8e12
        print(f"pass {rep} out = in + 1 GiB + {pad:>10d} B: fused {a:.4f} ms ({f(a):.3f})   bit-exact {b:.4f} ms ({f(b):.3f})", flush=True)
