cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
rm -f gpurun_out/r5/parity.jsonl
SAVGOL_PARITY_LOG=$PWD/gpurun_out/r5/parity.jsonl timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -60 > gpurun_out/r5/gputests.txt
python tools/parity_margins.py gpurun_out/r5/parity.jsonl > gpurun_out/r5/parity_margins.txt 2>&1
tail -25 gpurun_out/r5/gputests.txt
