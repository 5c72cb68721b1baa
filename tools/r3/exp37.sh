cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp37; rm -rf $O
rocprofv3 --kernel-trace --memory-copy-trace -d $O -o run --output-format csv -- python3 $R/tools/r3/exp37.py 2>&1 | grep median
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/exp37"
k = glob.glob(O + "/**/run_kernel_trace.csv", recursive=True); m = glob.glob(O + "/**/run_memory_copy_trace.csv", recursive=True)
ev = []
for r in csv.DictReader(open(k[0])): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:40]))
if m:
    rows = list(csv.DictReader(open(m[0])))
    for r in rows: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "?"))))
ev.sort()
for s, e, n in ev[-16:]:
    print(f"{(s - ev[-16][0]) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {n}")
PY
