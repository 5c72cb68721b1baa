cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so; F=tools/ab/lib_flat.so; G=SAVGOL_HIP_STREAM_DMA_GROUP
{ timeout 600 python -m pytest tests/test_gpu_stream.py -x -q -m gpu 2>&1 | tail -2
  echo "## n=16 fused: shipped (group by rule) / GROUP=128"; python tools/placement_stream.py $L $L@$G=128 --allocations 10 2>&1 | grep -v amdgpu.ids | tail -5
  for n in 4 6 8 11; do for fma in 1 0; do echo "## n=$n fma=$fma: shape table / flat (4,12) / shape table GROUP=64 / flat GROUP=64"; python tools/placement_stream.py $L $F $L@$G=64 $F@$G=64 --n $n --fma $fma --allocations 8 2>&1 | grep -v amdgpu.ids | tail -5; done; done
  echo "## n=16 bit-exact: GROUP 128 / 64 / 256"; python tools/placement_stream.py $L $L@$G=64 $L@$G=256 --n 16 --fma 0 --allocations 8 2>&1 | grep -v amdgpu.ids | tail -5
  echo "## n=24 fused (tap by tap): GROUP 128 / 64"; python tools/placement_stream.py $L $L@$G=64 --n 24 --fma 1 --allocations 8 2>&1 | grep -v amdgpu.ids | tail -5
} > gpurun_out/r5/placement_stream3.txt 2>&1
cat gpurun_out/r5/placement_stream3.txt
