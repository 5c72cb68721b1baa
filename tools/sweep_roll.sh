# A/B of the 2-D rolling kernel's work distribution (persistent vs one item per wave) and band count; config 4 shape, 128 frames
cd $GRAFT_REPO_ROOT
for one in 0 1; do for bands in 0 8 16 32 64; do
  echo "ONEWAVE=$one BANDS=$bands: $(SAVGOL_HIP_ROLL_ONEWAVE=$one SAVGOL_HIP_ROLL_BANDS=$bands python bench.py --workload image --images 128 --no-cpu --steps 3 --warmup 1 | grep -o '"avg_launch_ms": [0-9.]*')"
done; done
