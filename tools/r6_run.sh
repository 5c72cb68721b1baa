# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_2d.py tests/test_gpu_rccl_exchange.py tests/test_gpu_bench_contract.py -q -m gpu -x 2>&1 | tail -12 > gpurun_out/r6/gputests_rb.txt; grep -E "passed|failed" gpurun_out/r6/gputests_rb.txt
