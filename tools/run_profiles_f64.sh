cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py; O=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/r2_f64_sq -o run --output-format csv -- python3 $B --workload batch1d_f64 --c5-channels 1024 --no-cpu --steps 2 --warmup 1 > $O/r2_f64_sq.log 2>&1
tail -1 $O/r2_f64_sq.log | cut -c1-200
