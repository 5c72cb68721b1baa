# Scratch script of round 5's GPU calls (rewritten per call: `gpurun -- 'bash tools/r5_run.sh'`); this last form is the round's closing check --
# the GPU suite with every comparison logged, the smoke entry point, then the evidence run (tools/run_profiles_r5.sh; summarise on the CPU box with
# python tools/summarise_profiles_r5.py).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5
rm -f gpurun_out/r5/parity.jsonl
SAVGOL_PARITY_LOG=$PWD/gpurun_out/r5/parity.jsonl timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -30 > gpurun_out/r5/gputests.txt
python tools/parity_margins.py gpurun_out/r5/parity.jsonl > gpurun_out/r5/parity_margins.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5/smoke.txt 2>&1
[ -n "$SKIP_PROFILES" ] || bash tools/run_profiles_r5.sh > gpurun_out/r5_prof.log 2>&1
tail -4 gpurun_out/r5/gputests.txt; grep -c OVER gpurun_out/r5/parity_margins.txt; tail -1 gpurun_out/r5/smoke.txt
