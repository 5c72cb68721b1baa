# N > 1 plumbing of bench.py on a ONE-GPU box: every rank on device 0, gloo for the timing reduce (SAVGOL_BENCH_BACKEND / _DEVICE hooks).  Round 6: world 8
# for all five workloads, config 5's per-GPU chunking (several chunks per rank), ragged shards (4097 channels over 8 ranks), per-rank fields.  What
# it shows: rendezvous, sharding, barriers, max-over-ranks timing and the JSON line at N = 8 -- NOT a scaling curve (eight ranks share one GPU).
cd ${GRAFT_REPO_ROOT:-.}
export SAVGOL_BENCH_BACKEND=gloo SAVGOL_BENCH_DEVICE=0
W=${W:-8}
line() { grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    keep={k:d.get(k) for k in ('metric','value','unit','n_gpus','steps','ms_per_step','scaling','backend','rccl_ranks','exchange','error') if k in d}
    keep['roofline.frac']=(d.get('roofline') or {}).get('frac'); keep['config']=d.get('config'); keep['per_rank']=d.get('per_rank')
    print(json.dumps(keep))"; }
echo "== batch1d x$W (256 channels per rank)"; timeout 600 python bench.py --gpus $W --channels 256 --length 262144 --steps 2 --warmup 1 | line
echo "== batch1d x$W, 4097 channels split unevenly (513 + 7 x 512)"; timeout 600 python bench.py --gpus $W --total-channels 4097 --length 65536 --steps 2 --warmup 1 | line
echo "== f64 x$W (config 5's call: savgol_apply_batch_f64_tol, 8 channels per rank in chunks of 2)"; timeout 600 python bench.py --gpus $W --workload batch1d_f64 --c5-channels 8 --c5-chunk 2 --steps 2 --warmup 1 | line
echo "== stream x$W"; timeout 600 python bench.py --gpus $W --workload stream --streams 4096 --ticks 512 --steps 2 --warmup 1 | line
echo "== image x$W"; timeout 600 python bench.py --gpus $W --workload image --images 4 --size 1024 --steps 2 --warmup 1 | line
echo "== rowband x$W (torch exchange: ranks on one device cannot form an RCCL communicator)"; timeout 600 python bench.py --gpus $W --workload image --rowband --exchange torch --images 4 --size 2048 --steps 2 --warmup 1 2>&1 | line
echo "== rowband x2 --exchange c under the gloo hook: must print an error line and exit non-zero"; timeout 300 python bench.py --gpus 2 --workload image --rowband --exchange c --images 4 --size 1024 --steps 2 --warmup 1 2>&1 | grep '^{' | cut -c1-400; echo "exit code ${PIPESTATUS[0]}"
echo "== rowband x1 (ring of one through the C RCCL exchange)"; timeout 300 python bench.py --gpus 1 --workload image --rowband --images 16 --size 4096 --steps 2 --warmup 1 --no-cpu 2>&1 | line
