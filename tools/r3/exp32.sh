cd $GRAFT_REPO_ROOT
for lib in tools/ab/lib_pa4.so tools/ab/lib_pa8.so; do
  for pc in 4 3 2; do
    echo "== $lib SAVGOL_HIP_STREAM_PER_CU=$pc"
    SAVGOL_HIP_LIB=$GRAFT_REPO_ROOT/$lib SAVGOL_HIP_STREAM_PER_CU=$pc python tools/time_stream_block.py 2>&1 | grep -E "n=(16|17|24|32)" | sed 's/ per 4096.*= / /; s/ Gsamples.*//' | tr '\n' '|'; echo
  done
done
