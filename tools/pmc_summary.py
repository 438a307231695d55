#!/usr/bin/env python3
"""Mean per-dispatch counter values per kernel from rocprofv3 --pmc CSV output (one or more pass directories)."""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    n = max(len(v) for v in cs.values())
    if n < 3 or not any(t in k for t in ("k_conv", "k1_", "k_stem")):
        continue
    print(f"{k}   dispatches {n}")
    for c, v in sorted(cs.items()):
        v = v[len(v) // 4:]            # skip warm-up launches
        print(f"    {c:28s} {sum(v) / len(v):16.1f}")
