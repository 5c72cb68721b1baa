cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so; B=tools/ab/lib_before.so
{ timeout 1500 python -m pytest tests/test_gpu_2d.py -x -q -m gpu 2>&1 | tail -2
  for i in 1 2; do echo "## derivs, new"; python tools/time_2d_derivs.py 2>&1 | grep -v amdgpu.ids | grep -i "fused\|laplacian\|hessian\|gradient" | head -12; echo "## derivs, before"; SAVGOL_HIP_LIB=$PWD/$B python tools/time_2d_derivs.py 2>&1 | grep -v amdgpu.ids | grep -i "fused\|laplacian\|hessian\|gradient" | head -12; done
  for cfg in "5 4" "7 4" "3 5"; do set -- $cfg; echo "## n=$1 order $2 (general two-term smoothing): new / before"; python tools/placement_2d.py $L $B --allocations 5 --n $1 --order $2 --images 32 2>&1 | grep -v amdgpu.ids | tail -3 | head -1; done
} > gpurun_out/r5/xst_general.txt 2>&1
cat gpurun_out/r5/xst_general.txt
