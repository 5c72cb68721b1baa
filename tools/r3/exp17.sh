# round 3, GPU call 17: lock-stepped stream block push (WPB waves per block = neighbouring strips, barrier every BAR ticks, P+1 rows in flight)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp17; mkdir -p $O
for lib in savitzky-golay-filter_amd/lib/libsavgol_hip.so tools/ab/lib_sroll_w4.so tools/ab/lib_sroll_w8p7.so tools/ab/lib_sroll_w8b8p7.so tools/ab/lib_sroll_w16b8p7.so tools/ab/lib_sroll_w16b4p3.so; do
  for env in "X=0" "SAVGOL_HIP_STREAM_ONEWAVE=512"; do
    echo "== $lib $env"; env SAVGOL_HIP_LIB=$lib $env timeout 120 python tools/time_stream_block.py 2>&1 | grep -E "n=16 fma=1|n= 8 fma=1"
  done
done 2>&1 | tee $O/stream_lockstep.txt
