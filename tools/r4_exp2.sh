# round 4, GPU call: edge strips of the tile form on vector loads (SAVGOL_HIP_ROLL_TILE_EDGE=0: scalar path), 2 waves per block
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_exp2.txt; : > $O
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so
python -m pytest tests/test_gpu_2d.py tests/test_gpu_baseline_configs.py -x -q -m gpu 2>&1 | tail -5 | tee -a $O
for b in 1 0 2; do
echo "## n=7, 256 frames, boundary $b: tile / tile with scalar edge strips / strip walk" | tee -a $O
timeout 600 python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_TILE_EDGE=0 $L@SAVGOL_HIP_ROLL_TILE=0 --n 7 --images 256 --boundary $b 2>&1 | grep -v amdgpu.ids | tee -a $O
done
for n in 5 6; do
timeout 600 python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_TILE_EDGE=0 $L@SAVGOL_HIP_ROLL_TILE=0 --n $n --images 64 2>&1 | grep -v amdgpu.ids | tee -a $O
done
timeout 600 python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_TILE=0 --n 7 --images 64 2>&1 | grep -v amdgpu.ids | tee -a $O
timeout 600 python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_TILE=0 --n 7 --images 4 2>&1 | grep -v amdgpu.ids | tee -a $O
timeout 600 python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_TILE=0 --n 7 --images 1 2>&1 | grep -v amdgpu.ids | tee -a $O
