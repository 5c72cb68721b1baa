# 2-D additive kernel (n = 7): rows in flight x resident blocks per CU (LDS floor: 0 = as many as fit, 48 KB = 3 blocks, 72 = 2, 120 = 1)
cd $GRAFT_REPO_ROOT
L="tools/ab/lib_r_p3w3.so tools/ab/lib_r_p3w2.so tools/ab/lib_r_p5w2.so tools/ab/lib_r_p7w2.so tools/ab/lib_r_p9w2.so tools/ab/lib_r_p7w1.so tools/ab/lib_r_p11w1.so tools/ab/lib_r_p15w1.so"
for kb in 0 48 72 120; do
  echo "== SAVGOL_HIP_ROLL_LDS_KB=$kb"
  SAVGOL_HIP_ROLL_LDS_KB=$kb python tools/ab_2d.py $L --n 7 2>&1 | grep median
done
