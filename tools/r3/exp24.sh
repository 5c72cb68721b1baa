cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/exp24
rocprofv3 --kernel-trace -d $R/gpurun_out/exp24 -o run --output-format csv -- python3 $R/tools/r3/exp24.py > $R/gpurun_out/exp24.log 2>&1
python3 - <<'PY'
import csv, os, glob, collections
R = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(R + "/gpurun_out/exp24/**/run_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
prev_end = None
for r in rows:
    name = r["Kernel_Name"][:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    a = agg.setdefault(name, {"n": 0, "dur": 0, "gap": 0, "gaps": 0})
    a["n"] += 1; a["dur"] += e - s
    if prev_end is not None and prev_name == name: a["gap"] += s - prev_end; a["gaps"] += 1
    prev_end, prev_name = e, name
for k, a in agg.items():
    print(f"{k:60s} calls {a['n']:4d}  avg duration {a['dur']/a['n']/1e3:7.2f} us  avg gap to the previous launch of the same kernel {a['gap']/max(a['gaps'],1)/1e3:6.2f} us")
PY
