# PMC traffic of the 2-D rolling kernel on the full config 4 (512 frames of 4096^2, n=7, order 3, three modes per step)
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py; O=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r2_c4_fetch -o run --output-format csv -- python3 $B --workload image --no-cpu --steps 1 --warmup 1 > $O/r2_c4_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r2_c4_write -o run --output-format csv -- python3 $B --workload image --no-cpu --steps 1 --warmup 1 > $O/r2_c4_write.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/r2_c4_sq -o run --output-format csv -- python3 $B --workload image --no-cpu --steps 1 --warmup 1 > $O/r2_c4_sq.log 2>&1
tail -1 $O/r2_c4_sq.log | cut -c1-300
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/r2_c4_lds -o run --output-format csv -- python3 $B --workload image --no-cpu --steps 1 --warmup 1 > $O/r2_c4_lds.log 2>&1
