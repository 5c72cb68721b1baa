cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
{
echo "## default lib"; HALF_WINDOWS=16 python tools/time_stream_block.py 2>&1 | grep "n="
echo "## DMA off"; HALF_WINDOWS=16 SAVGOL_HIP_STREAM_DMA=0 python tools/time_stream_block.py 2>&1 | grep "n="
for cfg in "32 4 16 2" "32 4 8 2" "32 4 12 2" "32 8 8 2" "64 4 16 2" "64 4 8 2" "64 4 12 2" "48 4 8 2" "48 4 12 2" "96 4 12 2" "64 8 8 2" "64 2 12 2" "128 4 12 2" \
           "32 4 8 1" "32 4 12 1" "64 4 8 1" "64 4 12 1" "64 4 16 1" "96 4 12 1" "64 8 8 1" "48 4 12 1" "128 4 12 1" "64 2 12 1"; do set -- $cfg
  echo "## TR=$1 WPB=$2 PAIRS=$3 CHAINS=$4"; HALF_WINDOWS=16 SAVGOL_HIP_STREAM_DMA_TR=$1 SAVGOL_HIP_STREAM_DMA_WPB=$2 SAVGOL_HIP_STREAM_DMA_PAIRS=$3 SAVGOL_HIP_STREAM_DMA_CHAINS=$4 SAVGOL_HIP_LIB=$PWD/tools/ab/lib_dmaexp.so python tools/time_stream_block.py 2>&1 | grep "n=\|Error" | tail -3
done
} > gpurun_out/r5/stream_dma_4.txt 2>&1
tail -4 gpurun_out/r5/stream_dma_4.txt
