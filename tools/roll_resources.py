"""VGPRs / scratch of every sg2d_rolling_kernel instantiation, from `hipcc -S` output files (spill check after any change).
   python tools/roll_resources.py /tmp/rg0.s /tmp/rg1.s ..."""
import re, sys
rows = {}
for path in sys.argv[1:]:
    for line in open(path):
        m = re.match(r"\s*\.set _ZN2sg19sg2d_rolling_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELb([01])E\S*\.(num_vgpr|num_agpr|private_seg_size|numbered_sgpr), (\d+)", line)
        if m:
            rows.setdefault((int(m[1]), int(m[2]), int(m[3]), int(m[4]), int(m[5])), {})[m[6]] = int(m[7])
for (n, nt, nout, box, acc), r in sorted(rows.items()):
    flag = "  <-- SCRATCH" if r.get("private_seg_size", 0) else ""
    print(f"n={n:2d} nt={nt} nout={nout} box={box} acc={acc}: vgpr {r.get('num_vgpr'):3d} agpr {r.get('num_agpr', 0):3d} sgpr {r.get('numbered_sgpr'):3d} scratch {r.get('private_seg_size', 0)}{flag}")
