#!/usr/bin/env python3
"""Print one train step's kernel timeline (us) from a rocprofv3 kernel_trace.csv.
Usage: timeline.py kernel_trace.csv [anchor kernel substring] [which occurrence of the anchor]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2] if len(sys.argv) > 2 else "fm2_fwd"
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) - 2
s, e = idx[k], idx[k + 1]
t0 = int(rows[s]["Start_Timestamp"])
busy = 0
for r in rows[s:e]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += en - st
    print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f} q{r['Queue_Id']} {r['Kernel_Name'].replace('void ', '')[:90]}")
print(f"step span {(int(rows[e]['Start_Timestamp']) - t0) / 1e3:.1f} us, kernel busy {busy / 1e3:.1f} us, {e - s} launches")
