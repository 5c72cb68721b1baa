cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_1d.py -q -m gpu -k "many_short or every_half_window or golden or in_place" 2>&1 | tail -4 > gpurun_out/r5/tests1d.txt
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so
{ for n in 32 32 28 24; do python tools/ab_1d.py $L $L@SAVGOL_HIP_MOMENT_FORM=32 --n $n --rounds 20 2>&1 | grep -v amdgpu.ids | tail -3; done; python tools/ab_1d.py $L $L@SAVGOL_HIP_MOMENT_FORM=32 --n 32 --deriv 1 --rounds 10 2>&1 | tail -2; } > gpurun_out/r5/ab_momenth.txt 2>&1
tail -3 gpurun_out/r5/tests1d.txt; cat gpurun_out/r5/ab_momenth.txt
