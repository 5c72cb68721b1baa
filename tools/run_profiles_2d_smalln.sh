cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS -d $O/r2_sm_sq -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pmc_2d_smalln.py 2 > $O/r2_sm.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r2_sm_f -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pmc_2d_smalln.py 2 >> $O/r2_sm.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r2_sm_w -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pmc_2d_smalln.py 2 >> $O/r2_sm.log 2>&1
