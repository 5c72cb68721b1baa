# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests/test_gpu_2d.py tests/test_gpu_rccl_exchange.py tests/test_gpu_reference_programs.py -q -m gpu -x 2>&1 | tail -12 > gpurun_out/r6/gputests_rb.txt; grep -E "passed|failed" gpurun_out/r6/gputests_rb.txt
python bench.py > gpurun_out/r6/bench_rb.json 2> gpurun_out/r6/bench_rb.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6/bench_rb.json').read().strip().splitlines()[-1])
print(json.dumps(d['extra']['config4_rowband'])[:1500])
print(json.dumps(d['summary']))
PY
