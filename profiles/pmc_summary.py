#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel: pmc_summary.py counter_collection.csv [name filter]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if flt and flt not in k:
        continue
    acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "calls", len(next(iter(d.values()))))
