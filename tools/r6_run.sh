# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so
for n in 7 8 9 10 11 12 13; do timeout 300 python tools/ab_2d_gradient.py $L $L@SAVGOL_HIP_ROLL_TILE=0 --n $n 2>&1 | grep -v amdgpu; done > gpurun_out/r6/tiles_gradient_final.txt; cat gpurun_out/r6/tiles_gradient_final.txt
