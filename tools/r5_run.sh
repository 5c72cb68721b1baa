cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so
{ echo "## 1-D headline shape: chunks of 64 blocks (shipped) / old order / chunks of 256 / 32 / 1024"; python tools/placement_1d.py $L $L@SAVGOL_HIP_1D_XCD_CHUNK_LOG2=0 $L@SAVGOL_HIP_1D_XCD_CHUNK_LOG2=8 $L@SAVGOL_HIP_1D_XCD_CHUNK_LOG2=5 $L@SAVGOL_HIP_1D_XCD_CHUNK_LOG2=10 --channels 4096 --allocations 6 2>&1 | grep -v amdgpu.ids
  echo "## 2-D 64 x 4096^2 n=7: whole-frame chunks (shipped) / old order / 128 bands / 64 bands"; python tools/placement_2d.py $L $L@SAVGOL_HIP_ROLL_XCD_CHUNK_BANDS=0 $L@SAVGOL_HIP_ROLL_XCD_CHUNK_BANDS=128 $L@SAVGOL_HIP_ROLL_XCD_CHUNK_BANDS=64 --allocations 10 2>&1 | grep -v amdgpu.ids
  echo "## stream n=16 fused: group by rule (64, shipped) / GROUP=128 / GROUP=32 / MOMENT=0 / DMA=0"; python tools/placement_stream.py $L $L@SAVGOL_HIP_STREAM_DMA_GROUP=128 $L@SAVGOL_HIP_STREAM_DMA_GROUP=32 $L@SAVGOL_HIP_STREAM_MOMENT=0 $L@SAVGOL_HIP_STREAM_DMA=0 --allocations 12 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r5/placement_check.txt 2>&1
cat gpurun_out/r5/placement_check.txt
