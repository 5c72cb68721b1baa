cd $GRAFT_REPO_ROOT
export SAVGOL_BENCH_BACKEND=gloo SAVGOL_BENCH_DEVICE=0
echo "== batch1d x2"; timeout 300 python bench.py --gpus 2 --channels 256 --length 262144 --steps 2 --warmup 1 | cut -c1-400
echo "== f64 x2"; timeout 300 python bench.py --gpus 2 --workload batch1d_f64 --c5-channels 8 --c5-chunk 4 --steps 2 --warmup 1 | cut -c1-600
echo "== stream x2"; timeout 300 python bench.py --gpus 2 --workload stream --streams 4096 --ticks 512 --steps 2 --warmup 1 | cut -c1-400
echo "== image x2"; timeout 300 python bench.py --gpus 2 --workload image --images 4 --size 1024 --steps 2 --warmup 1 | cut -c1-400
echo "== rowband x2 (torch exchange: two ranks on one device cannot form an RCCL communicator)"; timeout 300 python bench.py --gpus 2 --workload image --rowband --exchange torch --images 4 --size 1024 --steps 2 --warmup 1 2>&1 | grep '^{' | cut -c1-700
echo "== rowband x2 --exchange c under the gloo hook: must print an error line and exit non-zero"; timeout 300 python bench.py --gpus 2 --workload image --rowband --exchange c --images 4 --size 1024 --steps 2 --warmup 1 2>&1 | grep '^{' | cut -c1-400; echo "exit code ${PIPESTATUS[0]}"
echo "== rowband x1"; timeout 300 python bench.py --gpus 1 --workload image --rowband --images 16 --size 4096 --steps 2 --warmup 1 --no-cpu 2>&1 | tail -1 | cut -c1-500
