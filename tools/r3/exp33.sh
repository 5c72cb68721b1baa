# 2-D additive kernel n = 7: the branch-free row loop (SG_ROLL_STRAIGHT: the compiler can count the memory operations in flight) x rows ahead
cd $GRAFT_REPO_ROOT
L="tools/ab/lib_r_s0p3w3.so tools/ab/lib_r_s1p3w3.so tools/ab/lib_r_s1p5w3.so tools/ab/lib_r_s1p7w3.so tools/ab/lib_r_s1p5w2.so tools/ab/lib_r_s1p7w2.so tools/ab/lib_r_s1p9w2.so tools/ab/lib_r_s1p11w2.so"
python tools/ab_2d.py $L --n 7 2>&1 | grep median
python tools/ab_2d.py $L --n 7 --images 256 2>&1 | grep median
python tools/ab_2d.py $L --n 6 2>&1 | grep median
