#!/usr/bin/env python3
"""Per-kernel averages of every counter found in rocprofv3 --pmc csv outputs under the given directories."""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(dict)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                per_dispatch[(row["Dispatch_Id"], row["Kernel_Name"])][row["Counter_Name"]] = float(row["Counter_Value"])
        for (_, k), cs in per_dispatch.items():
            for c, v in cs.items():
                acc[k][c].append(v)
for k, cs in sorted(acc.items()):
    if not any(s in k for s in ("fused", "sliced")):
        continue
    print(k[:110])
    for c, v in sorted(cs.items()):
        print(f"    {c:24s} n={len(v):3d} avg={sum(v) / len(v):.4g}")
