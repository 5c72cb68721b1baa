cd $GRAFT_REPO_ROOT
echo "plain below 2^23 (default):"; python tools/r3/exp38.py 2>&1 | tail -1
for m in 19 20 21 22; do echo "SAVGOL_HIP_PIPE_MIN_LOG2=$m (adaptive chunk):"; SAVGOL_HIP_PIPE_MIN_LOG2=$m python tools/r3/exp38.py 2>&1 | tail -1; done
for c in 17 18 19; do echo "SAVGOL_HIP_PIPE_MIN_LOG2=19 chunk 2^$c:"; SAVGOL_HIP_PIPE_MIN_LOG2=19 SAVGOL_HIP_PIPE_CHUNK_LOG2=$c python tools/r3/exp38.py 2>&1 | tail -1; done
