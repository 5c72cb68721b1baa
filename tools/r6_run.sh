# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
rm -f gpurun_out/r6/parity_all.log
SAVGOL_PARITY_LOG=gpurun_out/r6/parity_all.log timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r6/gputests_all.txt; grep -E "passed|failed|FAILED" gpurun_out/r6/gputests_all.txt
python tools/tick_offset_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r6/tick_probe.txt
