cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/exp19
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/exp19 -o run --output-format csv -- python3 $R/tools/r3/exp19.py > $R/gpurun_out/exp19.log 2>&1
python3 - <<'PY'
import csv, os, glob
R = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(R + "/gpurun_out/exp19/**/run_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "rolling" in r["Kernel_Name"]]
for r in rows:
    print(r["Kernel_Name"][:70], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "ms")
PY
