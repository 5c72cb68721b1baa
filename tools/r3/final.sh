# round 3, final GPU call: whole GPU suite, smoke, config 4's counter passes on the final 2-D sources, the driver-format line with traffic filled
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3_final; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
P=$GRAFT_REPO_ROOT/gpurun_out/r3_prof; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
SQ="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rm -rf $P/image_fetch $P/image_write $P/image_sq
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/image_fetch -o run --output-format csv -- python3 $B --workload image --no-cpu --steps 1 --warmup 1 > $P/image_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/image_write -o run --output-format csv -- python3 $B --workload image --no-cpu --steps 1 --warmup 1 > $P/image_write.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ -d $P/image_sq -o run --output-format csv -- python3 $B --workload image --no-cpu --steps 1 --warmup 1 > $P/image_sq.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarise_profiles_r3.py > $O/summarise.log 2>&1; tail -3 $O/summarise.log
python bench.py > $O/bench_line_final.json 2> $O/bench_line_final.err; echo "bench rc $?"
