cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu 2>&1 | tail -3
S=$(date +%s); python bench.py > gpurun_out/r5/bench_line.json 2> gpurun_out/r5/bench_line.err; echo "bench rc $? in $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_line.json').read().split('\n')[0])
e=d['extra']
print(d['value'], d['roofline']['frac'], d['roofline']['traffic'])
print('c3', e['config3']['block_push']['roofline']['frac'], e['config3']['block_push']['roofline']['traffic'], e['config3']['block_push_reference_order']['roofline_frac'])
print('c5', e['config5_slice']['roofline']['frac'], e['config5_slice']['opt_in_block_moments']['roofline']['frac'], e['config5_slice'].get('in_place'))
PY
