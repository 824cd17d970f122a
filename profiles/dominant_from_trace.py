#!/usr/bin/env python3
"""From a rocprofv3 kernel_trace.csv of `bench.py`: average duration of the dominant op's kernels
over (a) every launch and (b) the final back-to-back launches that `bench.py` times with HIP events
for its `roofline` entry (the last 30 launches of each kernel; in-step launches before them overlap
the next batch's index sort on another stream and last longer).
Usage: dominant_from_trace.py kernel_trace.csv [kernel substring ...]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
names = sys.argv[2:] or ["segreduce_chunk_kernel<4, 1, 1, true>", "segreduce_combine_kernel<4, 1, 1, true>"]
total_all = total_tail = 0.0
for n in names:
    d = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if n in r["Kernel_Name"]))
    dur = [x[1] for x in d]
    tail = dur[-30:]
    print(f"{n}: {len(dur)} launches, average {sum(dur) / len(dur) / 1e3:.1f} us; last {len(tail)} (back to back) average {sum(tail) / len(tail) / 1e3:.1f} us")
    total_all += sum(dur) / len(dur) / 1e3
    total_tail += sum(tail) / len(tail) / 1e3
print(f"op total: all launches {total_all:.1f} us, back-to-back launches {total_tail:.1f} us")
