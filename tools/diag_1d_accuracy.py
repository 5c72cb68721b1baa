"""Per filter: normwise error (against the double-accumulation oracle) of the default fp32 device kernel, of the plain 2n+1-tap kernel
(SAVGOL_HIP_OPT_PLAIN_SUMMATION) and of the reference's own fp32 arithmetic (the oracle's bit-exact restatement), on the signals of
tests/test_gpu_1d.py::test_fp32_kernels_against_the_reference_s_own_fp32_error.   python tools/diag_1d_accuracy.py 5 25 27"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from oracle import sgo
sg = ge.load_package()
L = sg.lib()
def normwise(a, b): return float(np.max(np.abs(a.astype(np.float64) - b)) / np.max(np.abs(b)))
for n in [int(v) for v in sys.argv[1:]] or [5, 25]:
    x = torch.empty((5, 40000 + 17 * n), dtype=torch.float32, device="cuda")
    sg.synth(x, channel0=3 * n)
    xh = x.cpu().numpy()
    rows = []
    for m in range(0, 7):
        for d in range(0, min(m, 2) + 1):
            for mode, dt in ((0, 1.0), (1, 1.0), (2, 0.5), (3, 1.0)):
                o = sgo.Filter(n, m, d, dt, mode)
                ref64 = o.apply_f64(xh.astype(np.float64))
                ref32 = o.apply(xh)
                f = sg.Filter(n, m, d, dt, mode)
                a = f.apply_tensor(x).cpu().numpy()
                L.savgol_hip_set_option(sg.SAVGOL_HIP_OPT_PLAIN_SUMMATION, 1)
                b = f.apply_tensor(x).cpu().numpy()
                L.savgol_hip_set_option(sg.SAVGOL_HIP_OPT_PLAIN_SUMMATION, 0)
                e, ep, er = normwise(a, ref64), normwise(b, ref64), normwise(ref32, ref64)
                # interior only (the edge outputs come from other code)
                ei = normwise(a[:, n:-n], ref64[:, n:-n]); eri = normwise(ref32[:, n:-n], ref64[:, n:-n])
                rows.append((e / max(er, 2.5e-7), m, d, mode, e, ep, er, ei, eri, float(np.max(np.abs(ref64))) / float(np.max(np.abs(xh)))))
    rows.sort(reverse=True)
    print(f"n={n}: ratio  m d mode   default     plain       reference   default(interior) reference(interior)  max|out|/max|in|")
    if os.environ.get("DIAG_MOMENT_RULE"):                 # only the filters the block-moment kernel keeps: poly_order >= 2, derivative <= 1
        rows = [r for r in rows if r[1] >= 2 and r[2] <= 1]
    for r in rows[:int(os.environ.get("DIAG_TOP", "8"))]:
        print(f"      {r[0]:5.2f}  {r[1]} {r[2]} {r[3]}     {r[4]:.3e}  {r[5]:.3e}  {r[6]:.3e}  {r[7]:.3e}  {r[8]:.3e}  {r[9]:.3g}")
