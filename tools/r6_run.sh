# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests/test_gpu_2d.py tests/test_gpu_reference_programs.py -q -m gpu -x 2>&1 | tail -12 > gpurun_out/r6/gputests_rb.txt; grep -E "passed|failed" gpurun_out/r6/gputests_rb.txt
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so; O=tools/ab/lib_dense_old.so
for s in "2 1" "7 3" "3 7" "12 5" "16 15"; do set -- $s; timeout 300 python tools/ab_2d_laplacian.py $L $O --nx $1 --ny $2 2>&1 | grep -v amdgpu; done > gpurun_out/r6/laplacian_rect_ab.txt; cat gpurun_out/r6/laplacian_rect_ab.txt
