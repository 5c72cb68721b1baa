cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_stream.py -x -q -m gpu 2>&1 | tail -3
for s in 1 2 3 4 5 6; do SAVGOL_FUZZ_SEED=$((s * 777)) SAVGOL_FUZZ_SCALE=4 timeout 900 python -m pytest tests/test_gpu_stream.py -x -q -m gpu -k "randomized" 2>&1 | tail -1; done
