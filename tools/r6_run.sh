# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
for keep in 256 512 1024 2048; do for tol in 1e-6 0; do echo "KEEP_MB=$keep tol=$tol"; SAVGOL_HIP_SCRATCH_KEEP_MB=$keep timeout 300 python tools/run_inplace.py --tol $tol --reps 4 2>&1 | grep "in place"; done; done > gpurun_out/r6/inplace_keep.txt; cat gpurun_out/r6/inplace_keep.txt
