cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so; B=tools/ab/lib_noxst.so
{ for n in 11 12 15 16; do echo "## n=$n (strip walk): x-stationary / folded"; python tools/placement_2d.py $L $B --allocations 5 --n $n --images 32 2>&1 | grep -v amdgpu.ids | tail -5; done
} > gpurun_out/r5/xst_walk.txt 2>&1
cat gpurun_out/r5/xst_walk.txt
