cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_exp7.txt; : > $O
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee -a $O
python tools/time_host_small.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee -a $O
