cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_2d.py -x -q -k "rolling_window_kernel_all_half_windows or rectangular or randomized or fused or derivative or hessian or gradient" 2>&1 | tail -8
timeout 600 python tools/sweep_perf.py 2d-orders 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sweep_orders_2pass.txt
