# the sample-ring kernel (tick loop unrolled 2n+1+P times) beyond n = 16, against the accumulator-ring kernel that serves those half windows
cd $GRAFT_REPO_ROOT
for lib in savitzky-golay-filter_amd/lib/libsavgol_hip.so tools/ab/lib_ring32_p3.so tools/ab/lib_ring32_p7.so; do
  for pc in 4 2; do
    echo "== $lib SAVGOL_HIP_STREAM_PER_CU=$pc"
    SAVGOL_HIP_LIB=$GRAFT_REPO_ROOT/$lib SAVGOL_HIP_STREAM_PER_CU=$pc python tools/time_stream_block.py 2>&1 | grep -E "n=(17|24|32)" | sed 's/ per 4096.*= / /; s/ Gsamples.*//' | tr '\n' '|'; echo
  done
done
