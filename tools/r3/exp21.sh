cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_1d.py tests/test_gpu_baseline_configs.py tests/test_gpu_reference_programs.py -x -q 2>&1 | tail -4
savitzky-golay-filter_amd/lib/time_batch_host tools/ab/lib_pre_edge.so savitzky-golay-filter_amd/lib/libsavgol_hip.so 2>&1 | tee gpurun_out/host_time_edge.txt
python tools/ab_1d_placements.py tools/ab/lib_pre_edge.so savitzky-golay-filter_amd/lib/libsavgol_hip.so --n 32 2>&1 | tail -3
python tools/ab_1d_placements.py tools/ab/lib_pre_edge.so savitzky-golay-filter_amd/lib/libsavgol_hip.so --n 32 --f64 --deriv 2 2>&1 | tail -3
python tools/ab_1d_placements.py tools/ab/lib_pre_edge.so savitzky-golay-filter_amd/lib/libsavgol_hip.so --n 8 2>&1 | tail -3
