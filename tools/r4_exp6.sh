# round 4, GPU call: the small fixes -- pool repro, RCCL exchange executed, scratch trim, overlap, service knobs; full-size tests of configs 3/4/5
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_exp6.txt; : > $O
timeout 300 tools/repro_null_stream_pool 2>&1 | tee gpurun_out/r04_null_stream_pool_repro.txt | tee -a $O
python -m pytest tests/test_gpu_rccl_exchange.py -x -q -m gpu 2>&1 | tail -15 | tee -a $O
python -m pytest tests/test_gpu_2d.py tests/test_gpu_1d.py -x -q -m gpu -k "overlapping or scratch_pool" 2>&1 | tail -15 | tee -a $O
python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_stream.py -x -q -m gpu -k "full_batch or full_slice or full_size" 2>&1 | tail -15 | tee -a $O
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee -a $O
