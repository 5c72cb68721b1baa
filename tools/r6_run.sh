# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
rm -f gpurun_out/r6/parity_all.log
SAVGOL_PARITY_LOG=gpurun_out/r6/parity_all.log timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 > gpurun_out/r6/gputests_all.txt; grep -E "passed|failed" gpurun_out/r6/gputests_all.txt
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so; O=tools/ab/lib_dense_old.so
for s in "2 1" "1 2" "4 2" "7 3" "3 7" "4 6" "12 5" "5 12" "16 1" "1 16" "16 15" "7 7" "3 3" "16 16"; do set -- $s; timeout 300 python tools/ab_2d.py $L $O --n $1 --ny $2 --method 1 --images 16 --order 2 2>&1 | grep -v amdgpu; done > gpurun_out/r6/dense_rect_ab.txt
ONLY=image bash tools/run_profiles_r6.sh > gpurun_out/r6/prof_image.log 2>&1
