cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so; B=tools/ab/lib_before.so
{ timeout 1200 python -m pytest tests/test_gpu_1d.py -x -q -m gpu 2>&1 | tail -2
  python tools/placement_1d.py $L $B --channels 4096 --allocations 6 2>&1 | grep -v amdgpu.ids
  python tools/placement_1d.py $L $B --channels 2048 --n 24 --allocations 6 2>&1 | grep -v amdgpu.ids | tail -6
} > gpurun_out/r5/momenth_b128.txt 2>&1
ONLY=f32 bash tools/run_profiles_r5.sh > gpurun_out/r5_prof_f32.log 2>&1
cat gpurun_out/r5/momenth_b128.txt
