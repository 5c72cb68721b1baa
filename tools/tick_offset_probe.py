"""the fused bank's PER-TICK kernel (and the walk, streams % 128 != 0) on streams with a large offset: python tools/tick_offset_probe.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
from oracle import sgo
from tests._util import normwise, fp32_bar
sg = load_package()
rng = np.random.default_rng(11)
for S, label in ((256, "per-tick kernel"), (200, "block push, walk (streams % 128 != 0)")):
    T = 400
    tt = np.arange(T)
    sb = np.sin(0.02 * tt)[:, None] * np.linspace(0.5, 1.5, S)[None, :] + rng.normal(0, 0.1, (T, S))
    for (n, m, d) in ((16, 2, 1), (16, 2, 2), (8, 3, 1)):
        row = []
        for off in (0.0, 10.0, 1000.0):
            x = (sb + off).astype(np.float32)
            bank = sg.StreamBank(S, n, m, d, 1.0, fma=True)
            dx = torch.from_numpy(x).cuda(); out = torch.zeros_like(dx)
            if S == 256:
                o1 = torch.zeros(S, device="cuda")
                for t in range(T):
                    bank.push(dx[t], o1); out[t] = o1
            else:
                bank.push_block(dx, T, out)
            torch.cuda.synchronize()
            o = sgo.Filter(n, m, d, 1.0, 0)
            xh = np.ascontiguousarray(x.T)
            hi = o.apply_f64(xh.astype(np.float64))[:, n:T - n]
            rb = sg.StreamBank(S, n, m, d, 1.0); want = torch.zeros_like(dx); rb.push_block(dx, T, want); torch.cuda.synchronize()
            ref = want.cpu().numpy()[2 * n:].T                                 # the reference's own stream arithmetic (one chain) = the bit-exact bank
            got = out.cpu().numpy()[2 * n:].T
            e, er = normwise(got, hi), normwise(ref, hi)
            row.append(f"off {off:g}: ours {e:.1e} ref {er:.1e} ({e / fp32_bar(er):.2f})")
        print(f"{label} n={n} m={m} d={d}: " + "   ".join(row), flush=True)
