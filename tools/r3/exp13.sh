# round 3, GPU call 13: one window buffer in the additive form, every half window (previous build vs this one, same process), then the 2-D suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp13; mkdir -p $O
for n in 1 2 3 4 5 6 7 8 9 10 12 14 16; do timeout 200 python tools/ab_2d.py tools/ab/lib_prev.so savitzky-golay-filter_amd/lib/libsavgol_hip.so --n $n 2>&1 | tail -2; done | tee $O/nb1_sweep.txt
timeout 1200 python -m pytest tests/test_gpu_2d.py -q -m gpu 2>&1 | tail -3
