#!/usr/bin/env python3
"""Print one train step's kernel timeline (us) from a rocprofv3 kernel_trace.csv."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "fm2_fwd"
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
s, e = idx[-2], idx[-1]
t0 = int(rows[s]["Start_Timestamp"])
busy = 0
for r in rows[s:e]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += en - st
    print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f}  {r['Kernel_Name'][:100]}")
print(f"step span {(int(rows[e]['Start_Timestamp']) - t0) / 1e3:.1f} us, kernel busy {busy / 1e3:.1f} us, {e - s} launches")
