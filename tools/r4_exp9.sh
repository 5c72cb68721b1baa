# round 4, GPU call: stream tile variants (streams per lane x rows per tile) on config 3's shape
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_exp9.txt; : > $O
for v in s4r16 s4r16w4 s4r24 s2r32 s2r48; do
echo "## $v" | tee -a $O
SAVGOL_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/lib_$v.so python tools/time_stream_block.py 2>&1 | grep -v amdgpu.ids | grep -E "n= 4|n= 8|n=16" | tee -a $O
done
echo "## walk" | tee -a $O
SAVGOL_HIP_STREAM_TILE=0 python tools/time_stream_block.py 2>&1 | grep -v amdgpu.ids | grep -E "n= 4|n= 8|n=16" | tee -a $O
