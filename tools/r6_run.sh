# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
rm -f gpurun_out/r6/parity_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SAVGOL_PARITY_LOG=gpurun_out/r6/parity_all.log timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r6/gputests_all.txt; grep -E "passed|failed|FAILED" gpurun_out/r6/gputests_all.txt
python tools/offset_probe_1d.py 2>&1 | grep -v amdgpu > gpurun_out/r6/offset_probe.txt
ONLY=stream bash tools/run_profiles_r6.sh > gpurun_out/r6/prof_stream.log 2>&1
cd "${GRAFT_REPO_ROOT:-.}"; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/bench_final.json 2> gpurun_out/r6/bench_final.err; tail -c 1300 gpurun_out/r6/bench_final.json
bash tools/soak_gpu.sh 6 4 1200 > gpurun_out/r6/soak_centre.txt 2>&1; grep -c passed gpurun_out/r6/soak_centre.txt; grep -i "failed" gpurun_out/r6/soak_centre.txt | head -3
