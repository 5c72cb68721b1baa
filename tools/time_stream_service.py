"""Per-tick wall latency of the resident tick service vs launch + synchronise per tick (config 3: 65 536 streams, n=16)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sg = load_package()
import torch
S, n, T = 65536, 16, 4000
x = torch.randn((256, S), device="cuda")
o = torch.zeros(S, device="cuda")
for host_bell in (0, 1):
    if host_bell:
        os.environ["SAVGOL_HIP_SERVICE_HOST_BELL"] = "1"
    bank = sg.StreamBank(S, n, 2, 1, 1e-3)
    torch.cuda.synchronize()
    if host_bell == 0:
        st = torch.cuda.current_stream()
        for t in range(100):
            bank.push(x[t % 256], o); st.synchronize()
        lat = []
        for t in range(T):
            t0 = time.perf_counter(); bank.push(x[t % 256], o); st.synchronize(); lat.append((time.perf_counter() - t0) * 1e6)
        lat = np.sort(lat)
        print(f"launch + stream synchronise per tick : p50 {lat[T//2]:.2f} us  p99 {lat[int(T*.99)]:.2f} us  min {lat[0]:.2f} us")
    bank.service_start(2000)
    for t in range(100):
        bank.service_tick(x[t % 256], o)
    lat = []
    for t in range(T):
        t0 = time.perf_counter(); rc = bank.service_tick(x[t % 256], o); lat.append((time.perf_counter() - t0) * 1e6)
        assert rc == 1
    bank.service_stop()
    lat = np.sort(lat)
    print(f"resident service, doorbell in {'pinned host memory' if host_bell else 'device memory (BAR)'} : p50 {lat[T//2]:.2f} us  p99 {lat[int(T*.99)]:.2f} us  min {lat[0]:.2f} us")
    break      # the mailbox choice is made once per process (static); run again with SAVGOL_HIP_SERVICE_HOST_BELL=1 for the other
