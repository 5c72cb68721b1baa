# SQ-side view of the same question: how long is a vector-memory instruction in flight (LEVEL / INSTS), how long an LDS one, and do the
# address / data FIFOs towards the texture unit fill up?  The 2-D additive kernel, the 1-D headline kernel, the bare walks and the flat copy.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp26; rm -rf $O; mkdir -p $O
try() { name=$1; prog=$2; shift; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o run --output-format csv -- $prog > $O/$name.log 2>&1; echo "$name rc $?"; }
for w in k m; do
  if [ $w = k ]; then P="python3 $R/tools/r3/exp26.py"; else P="$R/tools/membench_lockstep"; fi
  try ${w}_vmem "$P" SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  try ${w}_lds "$P" SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY
  try ${w}_fifo "$P" SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY
done
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT") + "/gpurun_out/exp26"
for name in ("k_vmem", "k_lds", "k_fifo", "m_vmem", "m_lds", "m_fifo"):
    fs = glob.glob(f"{O}/{name}/**/run_counter_collection.csv", recursive=True)
    if not fs:
        print(name, "no counters:", open(f"{O}/{name}.log").read()[-200:].replace("\n", " | ")); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        kn = r["Kernel_Name"]
        if not any(t in kn for t in ("rolling", "center", "k_flat", "k_rows<4, 4, 0>", "k_rows<8, 16, 8>", "k_rows<8, 4, 0>")): continue
        key = (kn.split("(")[0].replace("void ", "")[:44], r["Grid_Size"])
        agg.setdefault(key, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", name)
    for key, c in agg.items():
        v = {k: sum(x) / len(x) for k, x in c.items()}
        extra = ""
        if "SQ_INST_LEVEL_VMEM" in v: extra = f"  -> {v['SQ_INST_LEVEL_VMEM'] / max(v['SQ_INSTS_VMEM_RD'] + v['SQ_INSTS_VMEM_WR'], 1):.0f} cycles per vector-memory instruction"
        if "SQ_INST_LEVEL_LDS" in v and v.get("SQ_INSTS_LDS", 0) > 0: extra = f"  -> {v['SQ_INST_LEVEL_LDS'] / v['SQ_INSTS_LDS']:.0f} cycles per LDS instruction"
        print(f"  {key[0]:44s} grid {key[1]:>9s}: " + "  ".join(f"{k.replace('SQ_', '')}={x:.4g}" for k, x in v.items()) + extra)
PY
