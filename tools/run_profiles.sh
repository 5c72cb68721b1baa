set -x
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err; echo "bench rc $?"
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py; O=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --stats -d $O/r2_stats -o run --output-format csv -- python3 $B --no-cpu --no-extra > $O/r2_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r2_pmc_fetch -o run --output-format csv -- python3 $B --no-cpu --no-extra --steps 2 --warmup 1 > $O/r2_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r2_pmc_write -o run --output-format csv -- python3 $B --no-cpu --no-extra --steps 2 --warmup 1 > $O/r2_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/r2_pmc_sq -o run --output-format csv -- python3 $B --no-cpu --no-extra --steps 2 --warmup 1 > $O/r2_pmc_sq.log 2>&1
ls $O/r2_stats $O/r2_pmc_fetch
