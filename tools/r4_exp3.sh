# round 4, GPU call: the tile form on the other half windows (additive, n = 1..7) and on the general one- / two-term forms (variant builds)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_exp3.txt; : > $O
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so
A=tools/ab
python -m pytest tests/test_gpu_2d.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O
echo "## additive form, 64 frames CONSTANT: tile / strip walk" | tee -a $O
for n in 1 2 3 4 5 6 7; do
timeout 300 python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_TILE=0 --n $n 2>&1 | grep -v amdgpu.ids | tee -a $O
done
echo "## general forms (variant build with SG_ROLL_TILE_GENERAL=1): d=(1,0) rank 1, order 4 d=(0,0) rank 3 (walk), order 3 d=(2,0) rank 1, order 2 d=(1,1)" | tee -a $O
for n in 2 4 5; do
for cfg in "3 1 0" "3 2 0" "3 1 1" "4 1 0"; do set -- $cfg
timeout 300 python tools/ab_2d.py $A/lib_gen0.so $A/lib_gen0.so@SAVGOL_HIP_ROLL_TILE=0 --n $n --order $1 --dx $2 --dy $3 2>&1 | grep -v amdgpu.ids | tee -a $O
done; done
for n in 6 7; do
for cfg in "3 1 0" "3 2 0" "3 1 1" "4 1 0"; do set -- $cfg
timeout 300 python tools/ab_2d.py $A/lib_gen1.so $A/lib_gen1.so@SAVGOL_HIP_ROLL_TILE=0 --n $n --order $1 --dx $2 --dy $3 2>&1 | grep -v amdgpu.ids | tee -a $O
done; done
