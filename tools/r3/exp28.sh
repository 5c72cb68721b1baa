cd $GRAFT_REPO_ROOT
for lib in tools/ab/lib_sp3.so tools/ab/lib_sp5.so tools/ab/lib_sp7.so tools/ab/lib_sp9.so; do
  for pc in 4 3 2; do
    echo "== $lib SAVGOL_HIP_STREAM_PER_CU=$pc"
    SAVGOL_HIP_LIB=$GRAFT_REPO_ROOT/$lib SAVGOL_HIP_STREAM_PER_CU=$pc python tools/time_stream_block.py 2>&1 | grep "n=" | awk '{printf "%s %s %s ms | ", $1, $2, $3} END {print ""}'
  done
done
