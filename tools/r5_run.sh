cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
python bench.py > gpurun_out/r5/bench_line.json 2> gpurun_out/r5/bench_line.err; echo "bench rc $?"
