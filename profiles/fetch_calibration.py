#!/usr/bin/env python3
"""What does TCC FETCH_SIZE count for the access shape of these kernels?  rx_fm2_fwd (no bias table) with
UNIFORM indices over 26 equal fields of a 2 GiB table: every one of the B x 26 row reads names a distinct
row that no cache holds, so the bytes are known -- rows x row bytes as requested by the lanes, rows x 128
for D = 16 and D = 32 as lines moved (a 64-byte row is half of a 128-byte line).  Run under
`rocprofv3 --pmc FETCH_SIZE` (its own pass) and fold the counter file with --fold:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 profiles/fetch_calibration.py
    python3 profiles/fetch_calibration.py --fold out/*/*counter_collection.csv
"""
import os
import sys

B, NF, LAUNCHES = 65536, 26, 3

if len(sys.argv) > 2 and sys.argv[1] == "--fold":
    import csv
    vals = []
    for r in csv.DictReader(open(sys.argv[2])):
        if r["Counter_Name"] == "FETCH_SIZE" and "fm2_fwd_kernel" in r["Kernel_Name"]:
            vals.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    vals = [v for _, v in sorted(vals)]
    # launches come in order: LAUNCHES at D = 16, then LAUNCHES at D = 32 (the kernel's name does not carry D)
    for k, D in enumerate((16, 32)):
        mine = vals[k * LAUNCHES:(k + 1) * LAUNCHES]
        if not mine:
            continue
        kib = sum(mine) / len(mine)
        reads = B * NF
        req, lines, idx = reads * D * 4, reads * 128, reads * 8
        got = kib * 1024
        print(f"D = {D:3d}: FETCH_SIZE {got / 1e6:8.1f} MB per launch ({len(mine)} launches); requested row bytes {req / 1e6:7.1f} MB, "
              f"lines moved {lines / 1e6:7.1f} MB, index stream {idx / 1e6:5.1f} MB -> "
              f"factor to requested bytes {(req + idx) / got:5.2f}, factor to line bytes {(lines + idx) / got:5.2f}")
    sys.exit(0)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recsys_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
ops.set_index_check("deferred")
for D in (16, 32):
    rows = (2 << 30) // (D * 4)
    per = rows // NF
    base = [f * per for f in range(NF + 1)]
    table = torch.zeros(per * NF, D, device=dev)
    g = torch.Generator(device=dev).manual_seed(D)
    ex_base = torch.tensor(base[:NF - 1], device=dev)
    for i in range(LAUNCHES):
        sp = torch.randint(0, per, (B, NF), device=dev, generator=g)
        ops.fm2_fwd(table, None, sp[:, :NF - 1], ex_base, sp[:, NF - 1:], base[NF - 1], want_tsum=True)
    torch.cuda.synchronize()
    del table
ops.raise_if_index_error()
