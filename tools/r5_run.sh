cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
rm -f gpurun_out/r5/parity.jsonl
SAVGOL_PARITY_LOG=$PWD/gpurun_out/r5/parity.jsonl timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -30 > gpurun_out/r5/gputests.txt
python tools/parity_margins.py gpurun_out/r5/parity.jsonl > gpurun_out/r5/parity_margins.txt 2>&1
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so
{ for n in 32 18 20 22 23 24 16; do python tools/ab_1d.py $L $L@SAVGOL_HIP_MOMENT_FORM=32 --n $n --rounds 12 2>&1 | grep -v amdgpu.ids | tail -2; done; } > gpurun_out/r5/ab_momenth2.txt 2>&1
tail -6 gpurun_out/r5/gputests.txt; grep -c OVER gpurun_out/r5/parity_margins.txt; cat gpurun_out/r5/ab_momenth2.txt
