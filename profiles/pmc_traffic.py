#!/usr/bin/env python3
"""Folds two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command)
into profiles/pmc_traffic.json: per kernel, average per launch.

    python profiles/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv>

traffic_bytes = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB): on gfx950 the fetch counter
tallies each 128-byte read request as 64 bytes for 16-byte-per-lane loads, which is how every row
access of these kernels is issued; the write counter is exact for 16-byte-per-lane stores."""
import csv
import json
import os
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    if "at::" in k or "rocprim" in k:
        continue
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    out[short(k)] = {"fetch_kib": round(f, 1), "write_kib": round(w, 1), "traffic_bytes": round((2 * f + w) * 1024)}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (kernel_source_sha: bench.py refuses a file taken on other kernel sources)
out["_source_sha"] = bench.kernel_source_sha()
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
del out["_source_sha"]
for k, v in out.items():
    print(f"{k[:70]:70s} fetch {v['fetch_kib'] / 1024:9.2f} MiB  write {v['write_kib'] / 1024:9.2f} MiB")
