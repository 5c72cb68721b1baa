# round 3, GPU call 5: new tests (fused strided, per-call flags, interleaved rows, launch split, never-worse-than-reference, FMA bank), strided timings
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp5; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_1d.py tests/test_gpu_stream.py tests/test_gpu_2d.py -q -m gpu -k "fused_strided or per_call_flags or interleaved or many_short or never_worse or fma_bank or plain_summation or strided or additive or row_bands or tick_service" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -30 $O/pytest.log
timeout 300 python tools/time_strided.py > $O/strided.txt 2>&1; cat $O/strided.txt
