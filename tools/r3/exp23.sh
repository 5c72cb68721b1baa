cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -14
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
savitzky-golay-filter_amd/lib/time_batch_host tools/ab/lib_pre_edge.so savitzky-golay-filter_amd/lib/libsavgol_hip.so 2>&1 | tee gpurun_out/host_time_edge.txt
python tools/time_strided.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/strided_edge.txt | tail -12
