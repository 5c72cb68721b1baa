# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
timeout 600 python tools/time_2d_derivs.py > gpurun_out/r6/derivs.txt 2>&1
grep -v amdgpu gpurun_out/r6/derivs.txt
