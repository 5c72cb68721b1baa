# round 3, GPU call 1: correctness of the BOX / edge-strip 2-D kernels and the FMA stream bank, then A/B timings
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_2d.py tests/test_gpu_stream.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log
for combo in "1 1" "0 1" "1 0" "0 0"; do
  set -- $combo
  SAVGOL_HIP_ROLL_BOX=$1 SAVGOL_HIP_ROLL_EDGE=$2 timeout 300 python bench.py --workload image --no-cpu --steps 3 --warmup 1 > $O/image_box$1_edge$2.json 2> $O/image_box$1_edge$2.err
  python - <<PY
import json
d=json.loads(open("$O/image_box$1_edge$2.json").read().strip().splitlines()[-1])
print("BOX=$1 EDGE=$2", d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
done
for combo in "1 1" "0 0"; do
  set -- $combo
  for n in 2 4 7 8 12; do SAVGOL_HIP_ROLL_BOX=$1 SAVGOL_HIP_ROLL_EDGE=$2 timeout 120 python tools/ab_2d.py savitzky-golay-filter_amd/lib/libsavgol_hip.so --n $n 2>&1 | tail -1; done
done > $O/ab2d.txt 2>&1
cat $O/ab2d.txt
timeout 300 python tools/time_stream_block.py > $O/stream_block.txt 2>&1; cat $O/stream_block.txt
