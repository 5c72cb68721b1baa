# round 3, GPU call 6: accuracy sweep with the conditioning-aware bars, three-output Hessian, rectangular windows on the rolling kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp6; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_1d.py tests/test_gpu_2d.py -q -s -m gpu -k "own_fp32_error or plain_summation or rectangular or fused_gradient or randomized_derivative or rank4 or graph" > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "^n=|passed|failed|AssertionError" $O/pytest.log | head -60
timeout 300 python tools/time_2d_derivs.py > $O/derivs.txt 2>&1; cat $O/derivs.txt
timeout 300 python tools/sweep_perf.py 2d > $O/sweep2d.txt 2>&1; tail -20 $O/sweep2d.txt
