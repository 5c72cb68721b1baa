"""Times savgol_streambank_push_block for a few half windows (tools, not product)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sg = load_package()
import torch

S, T = int(os.environ.get("STREAMS", "65536")), int(os.environ.get("TICKS", "4096"))
x = torch.randn((T, S), device="cuda")
out = torch.empty_like(x)
for n, fma in [(n, f) for n in [int(v) for v in os.environ.get("HALF_WINDOWS", "4,8,16,17,24,32").split(",")] for f in (False, True)]:
    bank = sg.StreamBank(S, n, 2, 1, 1e-3, fma=fma)
    bank.push_block(x, T, out); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); bank.push_block(x, T, out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = min(ts)
    # sustained: seven launches back to back (what bench.py reports; the chip lowers its clock under the reference-order bank)
    K = 7
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        bank.push_block(x, T, out)
    e1.record(); torch.cuda.synchronize()
    sus = e0.elapsed_time(e1) / K
    print(f"n={n:2d} fma={int(fma)}: {ms:.3f} ms single / {sus:.3f} ms sustained per {T} ticks x {S} streams = {S*T/sus/1e6:.0f} Gsamples/s, "
          f"{8*S*T/sus/1e6:.0f} GB/s = {8*S*T/sus/1e6/8000:.3f} of 8 TB/s (single {8*S*T/ms/1e6/8000:.3f})")
