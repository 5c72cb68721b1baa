# round 3, GPU call 9: fp64 n=32 with half the multiply-adds as adds (timing only: what the symmetric-tap fold could buy); stream block push rows in flight; rect-window test
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp9; mkdir -p $O
timeout 300 python tools/ab_1d.py savitzky-golay-filter_amd/lib/libsavgol_hip.so tools/ab/lib_f64adds.so --f64 --n 32 --deriv 2 --channels 1024 --length 4194304 2>&1 | tail -4 | tee $O/f64adds.txt
for lib in savitzky-golay-filter_amd/lib/libsavgol_hip.so tools/ab/lib_sroll_p6.so tools/ab/lib_sroll_p9.so; do
  echo "== $lib"; SAVGOL_HIP_LIB=$lib timeout 200 python tools/time_stream_block.py 2>&1 | grep -E "n=16|n= 8 fma=1|n=32 fma=1"
done 2>&1 | tee $O/stream_p.txt
timeout 900 python -m pytest tests/test_gpu_2d.py -q -m gpu -k "rectangular or rank4" 2>&1 | tail -5
