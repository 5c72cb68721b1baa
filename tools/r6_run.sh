# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_1d.py tests/test_gpu_stream.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r6/gputests_1d.txt
E=tools/ab/lib_momexp.so
timeout 900 python tools/placement_stream.py $E $E@SAVGOL_HIP_STREAM_DMA_TR=48 $E@SAVGOL_HIP_STREAM_DMA_TR=64 $E@SAVGOL_HIP_STREAM_DMA_TR=64,SAVGOL_HIP_STREAM_DMA_PAIRS=12 \
   $E@SAVGOL_HIP_STREAM_DMA_TR=64,SAVGOL_HIP_STREAM_DMA_PAIRS=20 $E@SAVGOL_HIP_STREAM_DMA_TR=64,SAVGOL_HIP_STREAM_DMA_WPB=4,SAVGOL_HIP_STREAM_DMA_PAIRS=24 \
   $E@SAVGOL_HIP_STREAM_DMA_TR=64,SAVGOL_HIP_STREAM_DMA_WPB=4,SAVGOL_HIP_STREAM_DMA_PAIRS=32 $E@SAVGOL_HIP_STREAM_DMA_TR=96 $E@SAVGOL_HIP_STREAM_DMA_TR=128 \
   $E@SAVGOL_HIP_STREAM_DMA_TR=128,SAVGOL_HIP_STREAM_DMA_WPB=4,SAVGOL_HIP_STREAM_DMA_PAIRS=24 --allocations 8 > gpurun_out/r6/stream_tile_heights.txt 2>&1
timeout 600 python bench.py --workload batch1d_f64 --steps 3 --warmup 1 > gpurun_out/r6/bench_c5.json 2> gpurun_out/r6/bench_c5_err.txt
tail -4 gpurun_out/r6/gputests_1d.txt; cat gpurun_out/r6/stream_tile_heights.txt | tail -20; tail -c 600 gpurun_out/r6/bench_c5.json
