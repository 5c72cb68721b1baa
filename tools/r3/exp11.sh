# round 3, GPU call 11: the lock-stepped rolling kernel (WPB waves per block = neighbouring strips of one band, barrier every BAR rows, P+1 rows in flight)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp11; mkdir -p $O
SAVGOL_HIP_LIB=tools/ab/lib_w9b8p7.so timeout 900 python -m pytest tests/test_gpu_2d.py -q -m gpu -x -k "additive or all_half_windows or config4 or separable_method" 2>&1 | tail -4
LIBS="savitzky-golay-filter_amd/lib/libsavgol_hip.so tools/ab/lib_w9b8p7.so tools/ab/lib_w9b8p3.so tools/ab/lib_w6b8p7.so tools/ab/lib_w9b4p7.so tools/ab/lib_w9b16p7.so"
timeout 300 python tools/ab_2d.py $LIBS --n 7 2>&1 | tail -6 | tee $O/lockstep_n7.txt
timeout 300 python tools/ab_2d.py $LIBS --n 7 --images 256 2>&1 | tail -6 | tee -a $O/lockstep_n7.txt
timeout 300 python tools/ab_2d.py $LIBS --n 6 2>&1 | tail -6 | tee -a $O/lockstep_n7.txt
