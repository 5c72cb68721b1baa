# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
rm -f gpurun_out/r6/parity.jsonl
SAVGOL_PARITY_LOG=$PWD/gpurun_out/r6/parity.jsonl SAVGOL_PARITY_NOASSERT=1 timeout 2400 python -m pytest tests/test_gpu_2d.py -q -m gpu 2>&1 | tail -40 > gpurun_out/r6/gputests_2d.txt
python tools/parity_margins.py gpurun_out/r6/parity.jsonl > gpurun_out/r6/parity_margins_2d.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_1d.py -q -m gpu -k "plain_summation or own_fp32_error" 2>&1 | tail -15 > gpurun_out/r6/gputests_1d_sel.txt
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py -q -m gpu -x 2>&1 | tail -30 > gpurun_out/r6/bench_contract.txt
tail -5 gpurun_out/r6/gputests_2d.txt; grep -c OVER gpurun_out/r6/parity_margins_2d.txt; tail -3 gpurun_out/r6/gputests_1d_sel.txt; tail -5 gpurun_out/r6/bench_contract.txt
