# round 4: the three-chain 1-D kernels against round 3's one-chain build in one process, more rounds; accuracy tests again
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_exp5.txt; : > $O
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so
python -m pytest tests/test_gpu_1d.py -q -m gpu -x 2>&1 | tail -3 | tee -a $O
for n in 32 32 24 20; do
timeout 600 python tools/ab_1d.py $L tools/ab/lib_onechain.so --n $n --rounds 30 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O
done
timeout 600 python tools/ab_1d.py $L tools/ab/lib_onechain.so --n 32 --m 1 --rounds 20 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O
