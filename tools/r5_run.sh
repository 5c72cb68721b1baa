cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so
{ for i in 1 2 3 4 5 6 7 8; do python tools/ab_stream.py $L $L@SAVGOL_HIP_STREAM_MOMENT=0 $L@SAVGOL_HIP_STREAM_DMA=0 --n 16 --rounds 6 2>&1 | grep -v amdgpu.ids; done
  for i in 1 2 3; do python tools/ab_stream.py $L $L@SAVGOL_HIP_STREAM_DMA=0 --n 16 --fma 0 --rounds 6 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r5/stream_processes.txt 2>&1
cat gpurun_out/r5/stream_processes.txt | cut -c30-220
