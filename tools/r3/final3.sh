# round 3, after the two-pass rolling change (sg_2d_roll.hip / sg_2d.hpp / sg_2d.hip): whole suite + smoke, the 2-D counter passes again
# (config 4, Hessian, gradient), the driver-format line with every traffic field filled
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3_final3; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
P=$GRAFT_REPO_ROOT/gpurun_out/r3_prof; mkdir -p $P
D=$GRAFT_REPO_ROOT/gpurun_out/r3_derivs; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
SQ="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rm -rf $P/image_fetch $P/image_write $P/image_sq $D/fetch $D/write
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/image_fetch -o run --output-format csv -- python3 $B --workload image --no-cpu --steps 1 --warmup 1 > $P/image_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/image_write -o run --output-format csv -- python3 $B --workload image --no-cpu --steps 1 --warmup 1 > $P/image_write.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ -d $P/image_sq -o run --output-format csv -- python3 $B --workload image --no-cpu --steps 1 --warmup 1 > $P/image_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/fetch -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/time_2d_derivs.py > $D/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/write -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/time_2d_derivs.py > $D/write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarise_profiles_r3.py > $O/summarise.log 2>&1; tail -3 $O/summarise.log
python tools/pmc_summary.py --kernel 'sg2d_rolling_kernel<7, 1, 3, false, false>' --alg-bytes 17179869184 --fetch $D/fetch --write $D/write \
  --command 'rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 tools/time_2d_derivs.py (tools/r3/final3.sh)' \
  --workload 'savgol2d_hessian_batch_f32: 64 frames 4096x4096 fp32, n=7, order 3: Hxx, Hxy, Hyy from ONE rolling launch; algorithmic bytes = 4 B read + 3 x 4 B written per pixel' \
  --sources sg_2d_roll.hip sg_2d.hpp sg_2d.hip --out profiles/r03_2d_hessian_fused_pmc_summary.json; echo "hessian rc $?"
python tools/pmc_summary.py --kernel 'sg2d_rolling_kernel<7, 2, 2, false, false>' --alg-bytes 12884901888 --fetch $D/fetch --write $D/write \
  --command 'rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 tools/time_2d_derivs.py (tools/r3/final3.sh)' \
  --workload 'savgol2d_gradient_batch_f32: 64 frames 4096x4096 fp32, n=7, order 3: d/dx and d/dy from one launch; 4 B read + 2 x 4 B written per pixel' \
  --sources sg_2d_roll.hip sg_2d.hpp sg_2d.hip --out profiles/r03_2d_gradient_fused_pmc_summary.json; echo "gradient rc $?"
mkdir -p gpurun_out/r3_final3/profiles; cp profiles/r03_2d_*_pmc_summary.json gpurun_out/r3_final3/profiles/
python bench.py > $O/bench_line_final.json 2> $O/bench_line_final.err; echo "bench rc $?"
