# HBM traffic of the three-output Hessian launch and the two-output gradient launch (round 3): FETCH_SIZE / WRITE_SIZE passes of
# tools/time_2d_derivs.py, separate runs, --kernel-trace only.  Then: python tools/pmc_summary.py ... (see profiles/README.md)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3_derivs; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/time_2d_derivs.py > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/time_2d_derivs.py > $O/write.log 2>&1
tail -3 $O/write.log
