cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
python bench.py > gpurun_out/r5/bench_line_after.json 2> gpurun_out/r5/bench_line_after.err; echo "bench rc $?"
