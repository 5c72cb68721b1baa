cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_baseline_configs.py -x -q 2>&1 | tail -3
for i in 1 2; do python tools/time_stream_block.py 2>&1 | grep "n=" ; done
python bench.py --workload stream --no-cpu --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline'].get('median_launch_ms'))"
