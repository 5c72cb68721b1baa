cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
rm -f gpurun_out/r5/parity.jsonl
SAVGOL_PARITY_LOG=$PWD/gpurun_out/r5/parity.jsonl timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r5/gputests.txt
python tools/parity_margins.py gpurun_out/r5/parity.jsonl > gpurun_out/r5/parity_margins.txt 2>&1
HALF_WINDOWS=16,17,18,20,22,24,32 python tools/time_stream_block.py 2>&1 | grep "n=" > gpurun_out/r5/stream_dma_6.txt
tail -8 gpurun_out/r5/gputests.txt; cat gpurun_out/r5/stream_dma_6.txt | cut -c1-150
