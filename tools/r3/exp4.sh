# round 3, GPU call 4: suites after the host rewrite (plans, flags), row bands through the C ABI, host time per call r02 vs now, nt loads in the 2-D kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp4; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_2d.py tests/test_gpu_1d.py tests/test_gpu_stream.py tests/test_gpu_baseline_configs.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -25 $O/pytest.log
savitzky-golay-filter_amd/lib/rowband_demo > $O/rowband_demo.txt 2>&1; echo "rowband_demo rc $?"; tail -3 $O/rowband_demo.txt
savitzky-golay-filter_amd/lib/time_batch_host tools/ab/lib_r02.so savitzky-golay-filter_amd/lib/libsavgol_hip.so > $O/time_batch_host.txt 2>&1; cat $O/time_batch_host.txt
timeout 300 python tools/ab_2d.py savitzky-golay-filter_amd/lib/libsavgol_hip.so tools/ab/lib_ntload.so tools/ab/lib_r02.so --n 7 2>&1 | tail -3 | tee $O/ab_ntload.txt
