# round 4, GPU call: the tile form of the stream block push (config 3: 65 536 streams x 4096 ticks, n = 16) against the strip walk; strips per group
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_exp8.txt; : > $O
python -m pytest tests/test_gpu_stream.py -x -q -m gpu 2>&1 | tail -4 | tee -a $O
for v in "SAVGOL_HIP_STREAM_TILE=1" "SAVGOL_HIP_STREAM_TILE=0" "SAVGOL_HIP_STREAM_TILE_GROUP=8" "SAVGOL_HIP_STREAM_TILE_GROUP=16" "SAVGOL_HIP_STREAM_TILE_GROUP=64" "SAVGOL_HIP_STREAM_TILE_GROUP=256" "SAVGOL_HIP_STREAM_XCD=0"; do
echo "## $v" | tee -a $O
env $v python tools/time_stream_block.py 2>&1 | grep -v amdgpu.ids | grep -E "n= 4|n= 8|n=16" | tee -a $O
done
