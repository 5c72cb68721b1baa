# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests/test_gpu_bench_contract.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r6/gputests_bc.txt; grep -E "passed|failed" gpurun_out/r6/gputests_bc.txt
for i in 1 2; do python bench.py > gpurun_out/r6/bench_b2b_$i.json 2> gpurun_out/r6/bench_b2b_$i.err; tail -n 1 gpurun_out/r6/bench_b2b_$i.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
bp=d['extra']['config3']['block_push']; print(bp['ms'], bp['roofline']['frac'], bp['roofline'].get('copy_frac'), bp['roofline'].get('frac_of_copy'), bp['roofline'].get('placement_spread'))
print(json.dumps(d['summary']))"; done
python bench.py --workload stream --no-cpu > gpurun_out/r6/bench_b2b_stream.json 2>/dev/null; tail -n 1 gpurun_out/r6/bench_b2b_stream.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('copy_frac'), d['roofline'].get('frac_of_copy'))"
