# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_stream.py -q -m gpu 2>&1 | tail -4 > gpurun_out/r6/gputests_stream.txt
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so
for n in 16 8 24 32; do timeout 300 python tools/ab_stream.py $L tools/ab/lib_momexp.so --fma 0 --n $n --m 2 --d 1; done > gpurun_out/r6/stream_chunk8_ab.txt 2>&1
cat gpurun_out/r6/gputests_stream.txt | grep -E "passed|failed"; grep -v amdgpu gpurun_out/r6/stream_chunk8_ab.txt | tail -12
