# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
rm -f gpurun_out/r6/parity_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SAVGOL_PARITY_LOG=gpurun_out/r6/parity_all.log timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r6/gputests_all.txt; grep -E "passed|failed" gpurun_out/r6/gputests_all.txt
python tools/time_2d_derivs.py 2>&1 | grep -v amdgpu > gpurun_out/r6/derivs_final.txt; cat gpurun_out/r6/derivs_final.txt | tail -30
python tools/gx_margin_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r6/gx_probe_final.txt
ONLY=image bash tools/run_profiles_r6.sh > gpurun_out/r6/prof_image.log 2>&1
bash tools/soak_gpu.sh 6 4 1000 > gpurun_out/r6/soak_hf.txt 2>&1; grep -c passed gpurun_out/r6/soak_hf.txt; grep -i "failed" gpurun_out/r6/soak_hf.txt | head -3
