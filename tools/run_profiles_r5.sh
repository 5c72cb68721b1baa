# Round 5 evidence run (one gpurun call): the driver-format bench line, five fresh-process rocprofv3 kernel traces of the headline
# HISTORY: round 5's evidence run (--f64-moment is now the default config-5 call; see tools/run_profiles_r6.sh).
# (VERDICT r02 next #1a), and the FETCH / WRITE / SQ counter passes of the four dominant kernels (next #1b) -- counters in their own
# runs with --kernel-trace only, the program directly after `--`, as MI355X_MICROARCH.md prescribes.
#   gpurun --timeout 3000 -- 'bash tools/run_profiles_r5.sh'   then   python tools/summarise_profiles_r5.py
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5_prof; mkdir -p $O
# ONLY=stream (or f32 / f64 / image): just that kernel's counter passes (after a change to that kernel's sources)
if [ -z "$ONLY" ]; then
python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?"
fi
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
if [ -z "$ONLY" ]; then
for i in 1 2 3 4 5; do
  rocprofv3 --kernel-trace --stats -d $O/repro$i -o run --output-format csv -- python3 $B --no-cpu --no-extra > $O/repro$i.json 2> $O/repro$i.err
done
fi
SQ="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
pmc() {   # name, bench.py arguments...
  name=$1; shift
  if [ -n "$ONLY" ] && [ "$ONLY" != "$name" ]; then return; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${name}_fetch -o run --output-format csv -- python3 $B "$@" > $O/${name}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${name}_write -o run --output-format csv -- python3 $B "$@" > $O/${name}_write.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ -d $O/${name}_sq -o run --output-format csv -- python3 $B "$@" > $O/${name}_sq.log 2>&1
}
pmc f32 --no-cpu --no-extra --steps 2 --warmup 1
pmc f64 --workload batch1d_f64 --c5-channels 1024 --no-cpu --steps 2 --warmup 1
pmc f64m --workload batch1d_f64 --c5-channels 1024 --no-cpu --steps 2 --warmup 1 --f64-moment
pmc stream --workload stream --no-cpu --no-extra --steps 3 --warmup 1
pmc image --workload image --no-cpu --steps 1 --warmup 1
ls $O
