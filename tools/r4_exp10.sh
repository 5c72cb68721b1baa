# round 4: does the stream block push's fraction of the roofline depend on the call's length (launch ramp / tail of a 0.4 ms kernel)?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_exp10.txt; : > $O
timeout 120 tools/membench_tile2d 16 0 2>&1 | head -3 | tee -a $O
for T in 4096 8192 16384; do
for v in s4r16w4 s4r16 s2r32; do
echo "## ticks $T, $v" | tee -a $O
TICKS=$T HALF_WINDOWS=16 SAVGOL_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/lib_$v.so python tools/time_stream_block.py 2>&1 | grep -v amdgpu.ids | grep -E "n=16" | tee -a $O
done
echo "## ticks $T, walk" | tee -a $O
TICKS=$T HALF_WINDOWS=16 SAVGOL_HIP_STREAM_TILE=0 python tools/time_stream_block.py 2>&1 | grep -v amdgpu.ids | grep -E "n=16" | tee -a $O
done
