# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
rm -f gpurun_out/r6/parity_2d.log
SAVGOL_PARITY_LOG=gpurun_out/r6/parity_2d.log timeout 2400 python -m pytest tests/test_gpu_2d.py tests/test_gpu_baseline_configs.py -q -m gpu -x 2>&1 | tail -12 > gpurun_out/r6/gputests_2d.txt; grep -E "passed|failed" gpurun_out/r6/gputests_2d.txt
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so
for n in 11 12 13 14 15 16; do timeout 300 python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_TILE=0 --n $n 2>&1 | grep -v amdgpu; done > gpurun_out/r6/tiles_final.txt; cat gpurun_out/r6/tiles_final.txt
ONLY=image bash tools/run_profiles_r6.sh > gpurun_out/r6/prof_image.log 2>&1
