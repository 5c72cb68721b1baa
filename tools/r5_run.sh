cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_stream.py tests/test_gpu_reference_programs.py -q -m gpu 2>&1 | tail -4 > gpurun_out/r5/stream_tests2.txt
savitzky-golay-filter_amd/lib/c_api_demo > gpurun_out/r5/c_api_demo.txt 2>&1
{
echo "## default"; HALF_WINDOWS=16 python tools/time_stream_block.py 2>&1 | grep "n="
for v in ntst ntld ntboth; do echo "## $v"; HALF_WINDOWS=16 SAVGOL_HIP_LIB=$PWD/tools/ab/lib_$v.so python tools/time_stream_block.py 2>&1 | grep "n=\|Error" | tail -3; done
echo "## default again"; HALF_WINDOWS=16 python tools/time_stream_block.py 2>&1 | grep "n="
} > gpurun_out/r5/stream_nt.txt 2>&1
cat gpurun_out/r5/stream_tests2.txt; grep "stream bank" gpurun_out/r5/c_api_demo.txt | cut -c1-200; cat gpurun_out/r5/stream_nt.txt | cut -c1-150
