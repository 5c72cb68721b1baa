# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r6/gputests_all.txt; grep -E "passed|failed" gpurun_out/r6/gputests_all.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/bench_final.json 2> gpurun_out/r6/bench_final.err; tail -c 1300 gpurun_out/r6/bench_final.json
