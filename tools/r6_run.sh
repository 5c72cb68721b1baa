# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
for i in 1 2; do python bench.py > gpurun_out/r6/bench_w_$i.json 2> gpurun_out/r6/bench_w_$i.err; tail -n 1 gpurun_out/r6/bench_w_$i.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
bp=d['extra']['config3']['block_push']; print(bp['ms'], bp['roofline']['frac'], bp['roofline'].get('copy_frac'), bp['roofline'].get('frac_of_copy'), bp['roofline'].get('placement_spread'))
print(d['extra']['config3']['block_push_reference_order'].get('placement_spread'))
print(json.dumps(d['summary']))"; done
python bench.py --workload stream --no-cpu --no-extra > gpurun_out/r6/bench_w_stream.json 2>/dev/null; tail -n 1 gpurun_out/r6/bench_w_stream.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('copy_frac'), d['roofline'].get('frac_of_copy'), d['steps'], d['warmup'])"
