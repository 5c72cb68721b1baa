cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_stream.py -x -q 2>&1 | tail -3
python tools/time_stream_block.py 2>&1 | grep "n="
