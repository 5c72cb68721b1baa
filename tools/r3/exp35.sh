cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_2d.py tests/test_gpu_reference_programs.py -x -q 2>&1 | tail -3
python tools/sweep_perf.py 2d 2>&1 | grep -v amdgpu.ids | grep "n=" | awk '{print $1,$2,$3, "method 1:", $(NF-3), $(NF-2), $(NF-1), $NF}'
