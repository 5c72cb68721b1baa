# instruction-cache counters of the 2-D rolling kernel: n=7 (config 4, 128 frames) and n=2 for comparison
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/r2_ic_n7 -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pmc_2d_smalln.py 7 > $O/r2_ic_n7.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/r2_ic_n2 -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pmc_2d_smalln.py 2 > $O/r2_ic_n2.log 2>&1
tail -2 $O/r2_ic_n7.log | cut -c1-200
