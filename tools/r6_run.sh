# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_1d.py tests/test_gpu_2d.py tests/test_gpu_baseline_configs.py -q -m gpu 2>&1 | tail -15 > gpurun_out/r6/gputests_sel.txt
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so
for n in 13 14; do timeout 300 python tools/ab_2d.py $L tools/ab/lib_tr13a.so --n $n --images 64; done > gpurun_out/r6/tiles_n13_16.txt 2>&1
for n in 15 16; do timeout 300 python tools/ab_2d.py $L tools/ab/lib_tr13b.so --n $n --images 64; done >> gpurun_out/r6/tiles_n13_16.txt 2>&1
timeout 900 python bench.py > gpurun_out/r6/bench_line2.json 2> gpurun_out/r6/bench_err2.txt
tail -4 gpurun_out/r6/gputests_sel.txt; grep median gpurun_out/r6/tiles_n13_16.txt; python -c "
import json
d=json.loads(open('gpurun_out/r6/bench_line2.json').read().strip().splitlines()[-1])
print(json.dumps(d['summary']))
print(json.dumps(d['extra']['config5_slice'].get('in_place')), json.dumps(d['extra']['config5_slice'].get('in_place_exact_1e12')))"
