# round 3, after the edge rows moved into the tile kernels' launch (sg_k1d.hpp / sg_k1d_host.hpp / sg_api_1d.cpp): whole suite + smoke, five
# fresh-process traces of the headline, the 1-D counter passes again, the driver-format line with every traffic field filled
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3_final4; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
P=$GRAFT_REPO_ROOT/gpurun_out/r3_prof; rm -rf $P; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
for i in 1 2 3 4 5; do
  rocprofv3 --kernel-trace --stats -d $P/repro$i -o run --output-format csv -- python3 $B --no-cpu --no-extra > $P/repro$i.json 2> $P/repro$i.err
done
SQ="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
pmc() { name=$1; shift
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/${name}_fetch -o run --output-format csv -- python3 $B "$@" > $P/${name}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/${name}_write -o run --output-format csv -- python3 $B "$@" > $P/${name}_write.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ -d $P/${name}_sq -o run --output-format csv -- python3 $B "$@" > $P/${name}_sq.log 2>&1; }
pmc f32 --no-cpu --no-extra --steps 2 --warmup 1
pmc f64 --workload batch1d_f64 --c5-channels 1024 --no-cpu --steps 2 --warmup 1
cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_line_final.json 2> $O/bench_line_final.err; echo "bench rc $?"
cp $O/bench_line_final.json $P/bench_line.json
python tools/summarise_profiles_r3.py > $O/summarise.log 2>&1; tail -12 $O/summarise.log
mkdir -p $O/profiles; cp profiles/r03_headline_repro.json profiles/r03_bench_kernel_stats.csv profiles/r03_1d_f32_n32_pmc_summary.json profiles/r03_1d_f64_n32_pmc_summary.json profiles/r03_bench_line.json $O/profiles/
python bench.py > $O/bench_line_final2.json 2> $O/bench_line_final2.err; echo "bench rc $?"
