cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
{ for sk in 0 6 14; do for cfg in "4 12" "4 16" "4 24" "4 32" "2 24" "2 32" "8 12" "8 16"; do set -- $cfg
  echo "## taps computed $((33-2*sk-1+ (sk==0?1:0) )) WPB=$1 PAIRS=$2"
  SAVGOL_HIP_LIB=$PWD/tools/ab/lib_exp$sk.so SAVGOL_HIP_STREAM_DMA_TR=32 SAVGOL_HIP_STREAM_DMA_WPB=$1 SAVGOL_HIP_STREAM_DMA_PAIRS=$2 HALF_WINDOWS=16 python tools/time_stream_block.py 2>&1 | grep "fma=1"
done; done; } > gpurun_out/r5/stream_occ_x_taps.txt 2>&1
cat gpurun_out/r5/stream_occ_x_taps.txt | cut -c1-200
