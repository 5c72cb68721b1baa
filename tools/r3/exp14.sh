# round 3, GPU call 14: the new randomized tests (strided calls, row bands + rectangular windows), then the whole randomized set on two fresh seeds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp14; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x -k "randomized_strided or randomized_row_bands" > $O/pytest_new.log 2>&1; echo "rc $?"; tail -25 $O/pytest_new.log
bash tools/soak_gpu.sh 2 2 2>&1 | tee $O/soak.txt | tail -12
