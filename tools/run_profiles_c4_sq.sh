cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py; O=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d $O/r2_c4_sq2 -o run --output-format csv -- python3 $B --workload image --images 128 --no-cpu --steps 1 --warmup 1 > $O/r2_c4_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY -d $O/r2_c4_sq3 -o run --output-format csv -- python3 $B --workload image --images 128 --no-cpu --steps 1 --warmup 1 > $O/r2_c4_sq3.log 2>&1
tail -2 $O/r2_c4_sq3.log | cut -c1-200
