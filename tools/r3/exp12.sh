# round 3, GPU call 12: the additive n=7 kernel at 4 waves per SIMD (one window buffer instead of two, 1 or 3 rows ahead)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp12; mkdir -p $O
LIBS="savitzky-golay-filter_amd/lib/libsavgol_hip.so tools/ab/lib_p1nb1w4.so tools/ab/lib_p3nb1w4.so tools/ab/lib_p3nb1w3.so"
timeout 300 python tools/ab_2d.py $LIBS --n 7 2>&1 | tail -4 | tee $O/w4.txt
timeout 300 python tools/ab_2d.py $LIBS --n 7 --images 256 2>&1 | tail -4 | tee -a $O/w4.txt
timeout 300 python tools/ab_2d.py $LIBS --n 7 --boundary 0 2>&1 | tail -4 | tee -a $O/w4.txt
