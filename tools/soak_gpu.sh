# Soak: the randomized GPU parity tests (1-D both types, strided records, stream sequences, 2-D configurations, derivative frames,
# row bands / rectangular windows) on fresh seeds.
#   gpurun -- bash tools/soak_gpu.sh 5 3 [first]     (5 seeds starting at `first` (default 1), 3 x the committed iteration counts)
cd $GRAFT_REPO_ROOT
SEEDS=${1:-3}; SCALE=${2:-2}; FIRST=${3:-1}
for s in $(seq $FIRST $((FIRST + SEEDS - 1))); do
  echo "== seed offset $((s * 1000)), scale $SCALE"
  SAVGOL_FUZZ_SEED=$((s * 1000)) SAVGOL_FUZZ_SCALE=$SCALE python -m pytest tests -m gpu -x -q -k "randomized" 2>&1 | tail -4
done
