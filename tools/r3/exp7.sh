# round 3, GPU call 7: what limits the strip walk -- lockstep microbenchmark; block order / one item per wave in the real 2-D and stream kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp7; mkdir -p $O
timeout 600 tools/membench_lockstep > $O/lockstep.txt 2>&1; cat $O/lockstep.txt
for x in 1 0; do
  echo -n "2-D n=7 SAVGOL_HIP_ROLL_XCD=$x: "; SAVGOL_HIP_ROLL_XCD=$x timeout 120 python tools/ab_2d.py savitzky-golay-filter_amd/lib/libsavgol_hip.so --n 7 2>&1 | tail -1
  echo -n "2-D n=2 SAVGOL_HIP_ROLL_XCD=$x: "; SAVGOL_HIP_ROLL_XCD=$x timeout 120 python tools/ab_2d.py savitzky-golay-filter_amd/lib/libsavgol_hip.so --n 2 2>&1 | tail -1
done 2>&1 | tee $O/xcd2d.txt
for env in "SAVGOL_HIP_STREAM_XCD=1" "SAVGOL_HIP_STREAM_XCD=0" "SAVGOL_HIP_STREAM_ONEWAVE=256" "SAVGOL_HIP_STREAM_ONEWAVE=512" "SAVGOL_HIP_STREAM_ONEWAVE=1024" "SAVGOL_HIP_STREAM_ONEWAVE=512 SAVGOL_HIP_STREAM_XCD=0"; do
  echo "== $env"; env $env timeout 200 python tools/time_stream_block.py 2>&1 | grep -E "n=16|n= 4 fma=1|n=32 fma=1"
done 2>&1 | tee $O/stream_order.txt
