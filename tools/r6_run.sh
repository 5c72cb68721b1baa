# Scratch script of round 6's GPU calls (rewritten per call: `gpurun -- 'bash tools/r6_run.sh'`).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r6
O=$GRAFT_REPO_ROOT/gpurun_out/r6/rb_trace; rm -rf $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu > $O.json 2> $O.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
rows=[]
for p in glob.glob('gpurun_out/r6/rb_trace/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'strip_scatter' in r['Kernel_Name']]
print(len(rows), len(idx))
# the last 3 steps of the back-to-back run: print kernels from 40 before the last scatter
lo=max(0, idx[-1]-40)
t0=int(rows[lo]['Start_Timestamp'])
with open('gpurun_out/r6/rb_timeline.txt','w') as fh:
    for r in rows[lo:idx[-1]+2]:
        line="%9.1f %9.1f q%s %s" % ((int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, r.get('Queue_Id','?'), r['Kernel_Name'][:70])
        fh.write(line+"\n"); print(line)
PY
