cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_2d.py -m gpu -x -q 2>&1 | tail -4
python tools/time_2d_derivs.py 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r2_grad_fetch -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/time_2d_derivs.py > $O/r2_grad_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r2_grad_write -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/time_2d_derivs.py > $O/r2_grad_write.log 2>&1
