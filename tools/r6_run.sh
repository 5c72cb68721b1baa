# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
rm -f gpurun_out/r6/parity_all.log
SAVGOL_PARITY_LOG=gpurun_out/r6/parity_all.log timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r6/gputests_all.txt; grep -E "passed|failed" gpurun_out/r6/gputests_all.txt
bash tools/soak_gpu.sh 6 4 700 > gpurun_out/r6/soak_tiles.txt 2>&1; grep -c passed gpurun_out/r6/soak_tiles.txt; grep -i "failed\|error" gpurun_out/r6/soak_tiles.txt | head -5
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/bench_final.json 2> gpurun_out/r6/bench_final.err; tail -c 1300 gpurun_out/r6/bench_final.json
