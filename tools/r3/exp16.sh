# round 3, GPU call 16: the small-call service with the centre row in LDS and the single-stream command: host-path + stream parity, timings
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp16; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_1d.py tests/test_gpu_reference_programs.py tests/test_gpu_stream.py -q -m gpu -x -k "golden or reference_program or unit_test or in_place or leading_edge or matlab or threads or plain_c or boundary_aware or nan_and_inf or single_stream or scenarios" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.log
for s in 1 0; do echo "== SAVGOL_HIP_SMALL_SERVICE=$s"; SAVGOL_HIP_SMALL_SERVICE=$s timeout 200 python tools/time_host_small.py 2>&1 | head -5; done | tee $O/small_service.txt
for s in 1 0; do echo "== reference demo program (test_savgol_main: 360 points x 10000, n=6), SAVGOL_HIP_SMALL_SERVICE=$s"; SAVGOL_HIP_SMALL_SERVICE=$s timeout 200 oracle/_ref/test_savgol_main 2>&1 | grep -iE "throughput|Average time|PASS" | head -6; done | tee -a $O/small_service.txt
python - <<'PY' 2>&1 | tee -a $O/small_service.txt
import os, sys, time
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
sg = load_package()
import numpy as np
x = np.random.default_rng(1).normal(0, 1, 3000).astype(np.float32)
for n, label in ((16, "savgol_stream_push, n=16"), (5, "savgol_stream_push, n=5")):
    s = sg.Stream(n, 2, 1, 1e-3)
    for v in x[:100]: s.push(float(v))
    t0 = time.perf_counter()
    for v in x[100:2100]: s.push(float(v))
    print(f"{label}: {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us per sample (through ctypes), SAVGOL_HIP_SMALL_SERVICE={os.environ.get('SAVGOL_HIP_SMALL_SERVICE', '1')}")
PY
SAVGOL_HIP_SMALL_SERVICE=0 python - <<'PY' 2>&1 | tee -a $O/small_service.txt
import os, sys, time
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
sg = load_package()
import numpy as np
x = np.random.default_rng(1).normal(0, 1, 3000).astype(np.float32)
s = sg.Stream(16, 2, 1, 1e-3)
for v in x[:100]: s.push(float(v))
t0 = time.perf_counter()
for v in x[100:2100]: s.push(float(v))
print(f"savgol_stream_push, n=16: {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us per sample (through ctypes), SAVGOL_HIP_SMALL_SERVICE=0")
PY
