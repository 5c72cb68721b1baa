cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
for s in 203000 206000; do SAVGOL_FUZZ_SEED=$s SAVGOL_FUZZ_SCALE=4 timeout 900 python -m pytest tests/test_gpu_2d.py -x -q -m gpu -k "randomized" 2>&1 | tail -1; done
bash tools/soak_gpu.sh 10 4 300 > gpurun_out/r5/soak2.txt 2>&1
grep -c passed gpurun_out/r5/soak2.txt; grep -i "fail\|error" gpurun_out/r5/soak2.txt | head -5
