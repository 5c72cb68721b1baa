# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
rm -f gpurun_out/r6/parity_all.log
SAVGOL_PARITY_LOG=gpurun_out/r6/parity_all.log timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r6/gputests_all.txt
ONLY=image bash tools/run_profiles_r6.sh > gpurun_out/r6/prof_image.log 2>&1
cd "${GRAFT_REPO_ROOT:-.}"; tail -3 gpurun_out/r6/gputests_all.txt
