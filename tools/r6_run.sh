# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/bench_final.json 2> gpurun_out/r6/bench_final.err; tail -4 gpurun_out/r6/bench_final.err; tail -c 1400 gpurun_out/r6/bench_final.json
