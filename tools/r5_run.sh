cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_gpu_1d.py -x -q -m gpu -k "in_place or randomized" 2>&1 | tail -2
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, os
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
sg = load_package()
import torch, numpy as np
def t(fn, k=4):
    fn(); torch.cuda.synchronize()
    ts=[]
    for _ in range(k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts), max(ts)
for dtype, ch, L, n, m, d in (("f64", 1024, 1 << 22, 32, 4, 2), ("f32", 4096, 1 << 20, 32, 4, 0), ("f32", 4096, 1 << 20, 8, 3, 0)):
    tdt = torch.float64 if dtype == "f64" else torch.float32
    x = torch.randn((ch, L), dtype=tdt, device="cuda"); y = torch.empty_like(x)
    f = sg.Filter(n, m, d, 1.0, 0)
    a = t(lambda: f.apply_batch(x, y, ch, L, dtype=dtype, flags=0))
    b = t(lambda: f.apply_batch(y, y, ch, L, dtype=dtype, flags=0))
    print(f"{dtype} {ch} x {L} n={n}: out of place {a[0]:.3f}-{a[1]:.3f} ms, in place {b[0]:.3f}-{b[1]:.3f} ms (+{100*(b[0]/a[0]-1):.1f} %) each call synchronised")
    del x, y
PY
python bench.py --workload batch1d_f64 --no-cpu --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['roofline']['frac'], d.get('in_place'), d.get('extra',{}).get('in_place'))"
