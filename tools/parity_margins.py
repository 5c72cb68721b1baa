#!/usr/bin/env python3
"""Worst margin per test of every comparison tests/_util.check() logged (SAVGOL_PARITY_LOG=path python -m pytest tests -m gpu):
    python tools/parity_margins.py gpurun_out/r5/parity.jsonl
prints, per test function, the number of comparisons, the largest value / bar ratio, and every comparison whose bar is wider than 1e-6
(the cases where the REFERENCE's own fp32 error exceeds 1e-6) or that failed."""
import collections
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
by = collections.defaultdict(list)
for r in rows:
    by[r["test"].split("[")[0]].append(r)
print(f"{len(rows)} comparisons in {len(by)} tests")
for t, rs in sorted(by.items()):
    worst = max(rs, key=lambda r: r["value"] / r["bar"])
    wide = [r for r in rs if r["bar"] > 1.0000001e-6]
    fail = [r for r in rs if r["value"] >= r["bar"]]
    print(f"{t}: {len(rs)} comparisons, worst value/bar {worst['value'] / worst['bar']:.2f} ({worst['value']:.3g} vs {worst['bar']:.3g}, {worst['label']}); "
          f"{len(wide)} with a bar above 1e-6 (max {max((r['bar'] for r in wide), default=0):.3g}); {len(fail)} over their bar")
    for r in fail[:12]:
        print(f"    OVER  {r['value']:.3g} >= {r['bar']:.3g}  {r['label']}  [{r['test']}]")
