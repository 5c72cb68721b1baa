# round 4: counters of the 2-D tile kernel on config 4's shape (IMAGES frames of 4096^2, n=7, order 3, three modes per step)
cd /tmp && export TMPDIR=/tmp
IMAGES=${IMAGES:-512}
B=$GRAFT_REPO_ROOT/bench.py; O=$GRAFT_REPO_ROOT/gpurun_out
ARGS="--workload image --images $IMAGES --no-cpu --steps 1 --warmup 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r4_c4_fetch -o run --output-format csv -- python3 $B $ARGS > $O/r4_c4_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r4_c4_write -o run --output-format csv -- python3 $B $ARGS > $O/r4_c4_write.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $O/r4_c4_sq -o run --output-format csv -- python3 $B $ARGS > $O/r4_c4_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAVES -d $O/r4_c4_sq2 -o run --output-format csv -- python3 $B $ARGS > $O/r4_c4_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $O/r4_c4_tcc -o run --output-format csv -- python3 $B $ARGS > $O/r4_c4_tcc.log 2>&1
tail -1 $O/r4_c4_tcc.log | cut -c1-300
