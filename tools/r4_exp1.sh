# round 4, GPU call: the tile form of the additive n=7 kernel -- tile height, waves per block, resident blocks, what the edge strips cost
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_exp1.txt; : > $O
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so
A=tools/ab
echo "## tile rows / waves per block / edge strips skipped (timing only), 64 frames CONSTANT then 256 frames" | tee -a $O
timeout 600 python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_TILE=0 $A/lib_r12.so $A/lib_r20.so $A/lib_r24.so $A/lib_wpb1.so $A/lib_wpb2.so $A/lib_wpb8.so $A/lib_skipedge.so --n 7 2>&1 | grep -v amdgpu.ids | tee -a $O
timeout 600 python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_TILE=0 $A/lib_r12.so $A/lib_r20.so $A/lib_r24.so $A/lib_wpb1.so $A/lib_wpb2.so $A/lib_wpb8.so $A/lib_skipedge.so --n 7 --images 256 2>&1 | grep -v amdgpu.ids | tee -a $O
echo "## resident blocks per CU capped through dynamic LDS (SAVGOL_HIP_ROLL_TILE_CAP), 256 frames" | tee -a $O
timeout 600 python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_TILE_CAP=3 $L@SAVGOL_HIP_ROLL_TILE_CAP=2 $L@SAVGOL_HIP_ROLL_TILE_CAP=1 $A/lib_r24.so@SAVGOL_HIP_ROLL_TILE_CAP=2 $A/lib_r24.so@SAVGOL_HIP_ROLL_TILE_CAP=1 --n 7 --images 256 2>&1 | grep -v amdgpu.ids | tee -a $O
echo "## blocks in launch order (SAVGOL_HIP_ROLL_XCD=0)" | tee -a $O
timeout 600 python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_XCD=0 --n 7 --images 256 2>&1 | grep -v amdgpu.ids | tee -a $O
