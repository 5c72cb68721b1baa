cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so; B=tools/ab/lib_before.so
{ timeout 1500 python -m pytest tests/test_gpu_2d.py -x -q -m gpu 2>&1 | tail -2
  for n in 8 9 10; do for b in 0 2; do echo "## n=$n boundary $b: new heights / before"; python tools/placement_2d.py $L $B --allocations 5 --n $n --images 32 --boundary $b 2>&1 | grep -v amdgpu.ids | tail -3 | head -1; done; done
} > gpurun_out/r5/tile_rows_check.txt 2>&1
cat gpurun_out/r5/tile_rows_check.txt
