# round 4, GPU call: two partial sums per output in the fp32 1-D kernels -- error sweep (the test prints it) and time against the one-chain build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_exp4.txt; : > $O
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so
python -m pytest tests/test_gpu_1d.py -q -m gpu -k "reference_s_own_fp32_error" -s 2>&1 | grep -E "worst normwise|passed|failed|Error" | tee -a $O
echo "## one process, interleaved: two chains / one chain (round 3)" | tee -a $O
for n in 32 28 24 16 8; do
timeout 600 python tools/ab_1d.py $L tools/ab/lib_onechain.so --n $n 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O
done
timeout 600 python tools/ab_1d.py $L tools/ab/lib_onechain.so --n 32 --deriv 2 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O
python -m pytest tests/test_gpu_1d.py tests/test_gpu_stream.py -x -q -m gpu 2>&1 | tail -4 | tee -a $O
python bench.py --no-cpu 2>&1 | tail -1 | cut -c1-700 | tee -a $O
