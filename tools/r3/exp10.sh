# round 3, GPU call 10: the whole GPU suite on the current tree, smoke, the host-pointer call across lengths (zero-copy path), a full default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp10; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -12 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
for z in 0 16384 262144; do echo "== SAVGOL_HIP_ZERO_COPY_MAX=$z"; SAVGOL_HIP_ZERO_COPY_MAX=$z timeout 200 python tools/time_host_small.py 2>&1 | head -5; done | tee $O/host_small.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["roofline"].get("frac_at_median"), d["roofline"]["avg_launch_ms"], d.get("buffers"))
e=d["extra"]
print("c1", e["config1"]["device_resident"], e["config1"].get("reference_demo_360pt"))
print("c3", e["config3"]["block_push"]["roofline"]["frac"], e["config3"]["block_push_reference_order"]["roofline_frac"], e["config3"].get("from_c"))
print("c4", {k:v["roofline"]["frac"] for k,v in e["config4"]["modes"].items()})
print("c5", e["config5_slice"]["roofline"]["frac"], e["config5_slice"]["ms_per_pass"])
PY
