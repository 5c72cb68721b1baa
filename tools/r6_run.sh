# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
rm -f gpurun_out/r6/parity.jsonl
SAVGOL_PARITY_LOG=$PWD/gpurun_out/r6/parity.jsonl SAVGOL_PARITY_NOASSERT=1 timeout 2700 python -m pytest tests/test_gpu_2d.py -q -m gpu 2>&1 | tail -40 > gpurun_out/r6/gputests_2d.txt
python tools/parity_margins.py gpurun_out/r6/parity.jsonl > gpurun_out/r6/parity_margins_2d.txt 2>&1
timeout 600 python tools/time_2d_derivs.py > gpurun_out/r6/derivs.txt 2>&1
grep -E "passed|failed" gpurun_out/r6/gputests_2d.txt; grep -c OVER gpurun_out/r6/parity_margins_2d.txt; grep "OVER\|pass_order" gpurun_out/r6/parity_margins_2d.txt | head -20; grep "(2,0) method 2\|hessian" gpurun_out/r6/derivs.txt
