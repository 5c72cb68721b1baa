# what differs between a flat copy and the strip walk in the memory system?  rocprofv3 counters of tools/membench_lockstep (flat copy, free-running
# strip walks, lock-stepped whole-row walks): TLB misses, L2 request counts and latencies (LEVEL / REQ), stalls.  Counter passes only (--kernel-trace).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp25; mkdir -p $O
rocprofv3 -L > $O/counters_all.txt 2>&1
grep -o -E "\b(TCP_UTCL1|TCC_EA|TCC_TAG|TCC_HIT|TCC_MISS|TCC_REQ|TCC_BUBBLE|TCP_TCC|TCP_PENDING|TCP_TA|TCC_.*STALL|TCC_.*MALL|MALL)[A-Za-z0-9_]*" $O/counters_all.txt | sort -u > $O/counters.txt
wc -l $O/counters.txt
try() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o run --output-format csv -- $R/tools/membench_lockstep > $O/$name.log 2>&1; echo "$name rc $?"; }
try rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
try wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
try wrstall TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUBBLE_sum
try tcp TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
try tcpw TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT") + "/gpurun_out/exp25"
for name in ("rd", "wr", "wrstall", "tcp", "tcpw"):
    fs = glob.glob(f"{O}/{name}/**/run_counter_collection.csv", recursive=True)
    if not fs:
        print(name, "no counters collected:", open(f"{O}/{name}.log").read()[-300:].replace("\n", " | ")); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        key = (r["Kernel_Name"][:40], r["Grid_Size"])
        agg.setdefault(key, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", name)
    for key, c in agg.items():
        print(f"  {key[0]:40s} grid {key[1]:>9s}: " + "  ".join(f"{k}={sum(v)/len(v):.4g}" for k, v in c.items()))
PY
