# round 3, GPU call 2: what the 4096-column frame costs the strip walk -- other widths (strip quantisation, row pitch), both layouts
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp2; mkdir -p $O
for cols in 3840 4080 4096 4112 4320; do
  for edge in 0 1; do
    echo -n "cols=$cols edge=$edge box=1: "; SAVGOL_HIP_ROLL_BOX=1 SAVGOL_HIP_ROLL_EDGE=$edge timeout 120 python tools/ab_2d.py savitzky-golay-filter_amd/lib/libsavgol_hip.so --n 7 --cols $cols 2>&1 | tail -1
  done
done 2>&1 | tee $O/widths.txt
for bands in 8 16 32 64; do
  echo -n "bands=$bands: "; SAVGOL_HIP_ROLL_BANDS=$bands SAVGOL_HIP_ROLL_EDGE=0 timeout 120 python tools/ab_2d.py savitzky-golay-filter_amd/lib/libsavgol_hip.so --n 7 2>&1 | tail -1
done 2>&1 | tee $O/bands.txt
