cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
bash tools/soak_gpu.sh 8 3 90 > gpurun_out/r5/soak.txt 2>&1
bash tools/check_multirank_plumbing.sh > gpurun_out/r5/multirank.txt 2>&1
grep -c passed gpurun_out/r5/soak.txt; grep -i "fail\|error" gpurun_out/r5/soak.txt | head -5; grep '^==\|exit code' gpurun_out/r5/multirank.txt; grep -c '^{"metric"' gpurun_out/r5/multirank.txt
