"""Print the kernel timeline of the last few steps from a rocprofv3 --kernel-trace CSV:
start offset, duration, gap to the previous kernel's end on the same queue, kernel name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
mid = sys.argv[3] if len(sys.argv) > 3 else None        # optional: start at this fraction (percent) of the trace, or "name:K" = at the K-th kernel whose name contains `name`
if mid is None:
    tail = rows[-n:]
elif ":" in mid:
    name, k = mid.rsplit(":", 1)
    hits = [i for i, r in enumerate(rows) if name in r["Kernel_Name"]]
    tail = rows[hits[int(k)]:][:n]
else:
    tail = rows[len(rows) * int(mid) // 100:][:n]
t0 = int(tail[0]["Start_Timestamp"])
last_end = {}
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  gap {gap:7.2f}  q{q:>3}  {r['Kernel_Name'][:70]}")
