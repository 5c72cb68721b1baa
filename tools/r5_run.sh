cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so; B=tools/ab/lib_before.so
{ timeout 1500 python -m pytest tests/test_gpu_2d.py tests/test_gpu_baseline_configs.py -x -q -m gpu 2>&1 | tail -3
  echo "## n=7 CONSTANT: x-stationary / before"; python tools/placement_2d.py $L $B --allocations 8 2>&1 | grep -v amdgpu.ids | tail -6
  echo "## n=7 VALID"; python tools/placement_2d.py $L $B --allocations 6 --boundary 0 2>&1 | grep -v amdgpu.ids | tail -5
  for n in 3 5 9 10; do echo "## n=$n"; python tools/placement_2d.py $L $B --allocations 5 --n $n 2>&1 | grep -v amdgpu.ids | tail -5; done
} > gpurun_out/r5/xst.txt 2>&1
cat gpurun_out/r5/xst.txt
