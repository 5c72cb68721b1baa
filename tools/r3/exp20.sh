cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_2d.py tests/test_gpu_reference_programs.py tests/test_gpu_baseline_configs.py -x -q 2>&1 | tail -5
timeout 600 python tools/sweep_perf.py 2d-orders 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sweep_orders_2pass.txt
SAVGOL_HIP_ROLL_SPLIT=0 timeout 600 python tools/sweep_perf.py 2d-orders 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sweep_orders_nosplit.txt
