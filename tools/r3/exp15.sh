# round 3, GPU call 15: the small-call service -- parity of every host-pointer test, the reference's own programs, timing on / off; Hessian traffic
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp15; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_1d.py tests/test_gpu_reference_programs.py tests/test_gpu_stream.py -q -m gpu -x -k "golden or reference_program or unit_test or in_place or leading_edge or matlab or threads or plain_c or row_bands_from or boundary_aware or nan_and_inf or single_stream" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.log
for s in 1 0; do echo "== SAVGOL_HIP_SMALL_SERVICE=$s"; SAVGOL_HIP_SMALL_SERVICE=$s timeout 200 python tools/time_host_small.py 2>&1 | head -4; done | tee $O/small_service.txt
for s in 1 0; do echo "== reference demo program, SAVGOL_HIP_SMALL_SERVICE=$s"; SAVGOL_HIP_SMALL_SERVICE=$s timeout 200 oracle/_ref/test_savgol_main 2>&1 | grep -iE "throughput|Msamples|time|PASS" | head -6; done | tee -a $O/small_service.txt
bash tools/run_profiles_2d_r3.sh
