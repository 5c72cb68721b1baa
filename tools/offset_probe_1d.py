"""1-D batch fp32 default kernels and the fused stream bank on data whose OFFSET dwarfs the signal: distance from the double oracle against the reference's
own fp32 distance (the rule: ours <= max(1e-6, 1.1 x ref)).  Derivative filters sum to zero: does the block-moment arithmetic keep up with the reference?
   python tools/offset_probe_1d.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
from oracle import sgo
from tests._util import normwise, fp32_bar
sg = load_package()
rng = np.random.default_rng(5)
L = 20000
t = np.arange(L)
base = np.sin(0.01 * t)[None, :] * np.linspace(0.5, 1.5, 4)[:, None] + rng.normal(0, 0.1, (4, L))
print("== 1-D batch fp32 (POLYNOMIAL mode, interior samples compared)")
for n, m in ((8, 3), (16, 2), (25, 4), (32, 4)):
    for d in (0, 1, 2):
        row = []
        for off in (0.0, 10.0, 1000.0):
            x = (base + off).astype(np.float32)
            f = sg.Filter(n, m, d, 1.0, 0); o = sgo.Filter(n, m, d, 1.0, 0)
            got = f.apply_tensor(torch.from_numpy(x).cuda()).cpu().numpy()[:, n:L - n]
            hi = o.apply_f64(x)[:, n:L - n]; ref = o.apply(x)[:, n:L - n]
            e, er = normwise(got, hi), normwise(ref, hi)
            row.append(f"off {off:g}: ours {e:.1e} ref {er:.1e} ({e / fp32_bar(er):.2f})")
        print(f"n={n} m={m} d={d}: " + "   ".join(row), flush=True)
print("== stream bank, fused multiply-add form (block push), n=16 m=2 d=1 and n=16 m=2 d=0")
S, T = 256, 2048
tt = np.arange(T)
sb = np.sin(0.02 * tt)[:, None] * np.linspace(0.5, 1.5, S)[None, :] + rng.normal(0, 0.1, (T, S))
for d in (0, 1, 2):
    row = []
    for off in (0.0, 10.0, 1000.0):
        x = (sb + off).astype(np.float32)
        bank = sg.StreamBank(S, 16, 2, d, 1.0, fma=True)
        dx = torch.from_numpy(x).cuda(); out = torch.zeros_like(dx)
        bank.push_block(dx, T, out); torch.cuda.synchronize()
        o = sgo.Filter(16, 2, d, 1.0, 0)
        hi = o.apply_f64(x.T.astype(np.float64).copy())[:, 16:T - 16]          # centre outputs j = 16 .. T-17 arrive at ticks 32 .. T-1
        rb = sg.StreamBank(S, 16, 2, d, 1.0); want = torch.zeros_like(dx); rb.push_block(dx, T, want); torch.cuda.synchronize()
        ref = want.cpu().numpy()[32:].T                                        # the reference's own stream arithmetic (one chain) = the bit-exact bank
        got = out.cpu().numpy()[32:].T
        e, er = normwise(got, hi), normwise(ref, hi)
        row.append(f"off {off:g}: ours {e:.1e} ref {er:.1e} ({e / fp32_bar(er):.2f})")
    print(f"d={d}: " + "   ".join(row), flush=True)
