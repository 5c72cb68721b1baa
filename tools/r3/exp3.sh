# round 3, GPU call 3: 2-D tests after the box form was tied to frame rows; band heights below 64 rows; 1-D + stream suites after the host rewrite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp3; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_2d.py tests/test_gpu_1d.py tests/test_gpu_stream.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log
for bands in 0 64 128 256; do
  echo -n "bands=$bands: "; SAVGOL_HIP_ROLL_BANDS=$bands timeout 120 python tools/ab_2d.py savitzky-golay-filter_amd/lib/libsavgol_hip.so --n 7 2>&1 | tail -1
done 2>&1 | tee $O/bands.txt
timeout 300 python bench.py --workload image --no-cpu --steps 3 --warmup 1 > $O/image.json 2> $O/image.err; tail -c 600 $O/image.json
