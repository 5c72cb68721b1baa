cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_2d.py tests/test_gpu_reference_programs.py -x -q 2>&1 | tail -2
for lib in tools/ab/lib_dense_before.so savitzky-golay-filter_amd/lib/libsavgol_hip.so; do
  echo "== $lib"
  SAVGOL_HIP_LIB=$GRAFT_REPO_ROOT/$lib python tools/sweep_perf.py 2d 2>&1 | grep -v amdgpu.ids | grep "n=" | awk '{printf "%s%s %s | ", $2, $3, $(NF-1)} END {print ""}'
done
