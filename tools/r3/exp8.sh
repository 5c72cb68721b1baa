# round 3, GPU call 8: rows in flight / waves per SIMD of the additive n=7 kernel, with block order and the persistent grid
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp8; mkdir -p $O
LIBS="savitzky-golay-filter_amd/lib/libsavgol_hip.so tools/ab/lib_p3w2.so tools/ab/lib_p5w3.so tools/ab/lib_p5w2.so tools/ab/lib_p7w2.so"
for env in "X=1" "SAVGOL_HIP_ROLL_XCD=0" "SAVGOL_HIP_ROLL_ONEWAVE=0" "SAVGOL_HIP_ROLL_ONEWAVE=0 SAVGOL_HIP_ROLL_XCD=0"; do
  echo "== $env"; env $env timeout 300 python tools/ab_2d.py $LIBS --n 7 2>&1 | tail -5
done 2>&1 | tee $O/variants.txt
