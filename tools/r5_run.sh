cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
{ for e in SAVGOL_HIP_STREAM_MOMENT=0 SAVGOL_HIP_STREAM_DMA=0; do echo "## $e: stream tests"; env $e timeout 900 python -m pytest tests/test_gpu_stream.py -x -q -m gpu 2>&1 | tail -1; done
  for e in SAVGOL_HIP_1D_XCD_CHUNK_LOG2=0 SAVGOL_HIP_1D_XCD_CHUNK_LOG2=3 SAVGOL_HIP_1D_XCD_CHUNK_LOG2=40 SAVGOL_HIP_MOMENT_FORM=32; do echo "## $e: 1-D tests"; env $e timeout 1500 python -m pytest tests/test_gpu_1d.py -x -q -m gpu 2>&1 | tail -1; done
  for e in SAVGOL_HIP_ROLL_XCD_CHUNK_BANDS=0 SAVGOL_HIP_ROLL_XCD_CHUNK_BANDS=37 SAVGOL_HIP_ROLL_TILE=0 SAVGOL_HIP_ROLL_BOX=0; do echo "## $e: 2-D tests"; env $e timeout 1500 python -m pytest tests/test_gpu_2d.py -x -q -m gpu 2>&1 | tail -1; done
} > gpurun_out/r5/switch_matrix.txt 2>&1
cat gpurun_out/r5/switch_matrix.txt
