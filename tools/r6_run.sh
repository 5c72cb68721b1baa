# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
ONLY=image bash tools/run_profiles_r6.sh > gpurun_out/r6_prof_image.log 2>&1
cd "${GRAFT_REPO_ROOT:-.}"
rm -f gpurun_out/r6/parity.jsonl
SAVGOL_PARITY_LOG=$PWD/gpurun_out/r6/parity.jsonl timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r6/gputests.txt
python tools/parity_margins.py gpurun_out/r6/parity.jsonl > gpurun_out/r6/parity_margins.txt 2>&1
grep -E "passed|failed" gpurun_out/r6/gputests.txt; grep -c OVER gpurun_out/r6/parity_margins.txt; head -1 gpurun_out/r6/parity_margins.txt; ls gpurun_out/r6_prof | grep image
