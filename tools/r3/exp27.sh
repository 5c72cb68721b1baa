# stream block push: fewer resident blocks per CU x more rows in flight (the bare walk liked 4 waves per CU with 8 rows in flight: 0.72)
cd $GRAFT_REPO_ROOT
for lib in tools/ab/lib_sp3.so tools/ab/lib_sp7.so tools/ab/lib_sp11.so; do
  for pc in 4 2 1; do
    echo "== $lib SAVGOL_HIP_STREAM_PER_CU=$pc"
    SAVGOL_HIP_LIB=$GRAFT_REPO_ROOT/$lib SAVGOL_HIP_STREAM_PER_CU=$pc python tools/time_stream_block.py 2>&1 | grep -E "n=16|n= 4 fma=1"
  done
done
