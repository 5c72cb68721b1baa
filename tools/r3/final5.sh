# round 3, after the stream block-push retune (sg_stream_roll.hip): suite + smoke, the stream counter passes, the driver-format line
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3_final5; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
P=$GRAFT_REPO_ROOT/gpurun_out/r3_prof; rm -rf $P; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
SQ="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
pmc() { name=$1; shift
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/${name}_fetch -o run --output-format csv -- python3 $B "$@" > $P/${name}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/${name}_write -o run --output-format csv -- python3 $B "$@" > $P/${name}_write.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ -d $P/${name}_sq -o run --output-format csv -- python3 $B "$@" > $P/${name}_sq.log 2>&1; }
pmc stream --workload stream --no-cpu --no-extra --steps 3 --warmup 1
cd $GRAFT_REPO_ROOT
python tools/summarise_profiles_r3.py > $O/summarise.log 2>&1; tail -4 $O/summarise.log
mkdir -p $O/profiles; cp profiles/r03_stream_block_pmc_summary.json $O/profiles/
python bench.py > $O/bench_line_final.json 2> $O/bench_line_final.err; echo "bench rc $?"
python tools/time_stream_block.py 2>&1 | grep "n=" > $O/stream_block.txt
