#!/usr/bin/env python3
"""How many independent row reads per second does the card deliver, by row size and by table
footprint?  rx_fm2_fwd without the bias table on 26 equal fields with UNIFORM indices (so nearly every
read is a distinct row and no cache holds it): B x 26 row reads per launch, B x (D + 1) floats out.
Prints reads/s and bytes/s for every (D, rows) pair; the launch is re-issued back to back between two
HIP events.  Usage: access_probe.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recsys_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
NF = 26
REPS = 20
dev = torch.device("cuda", 0)
ops.set_index_check("deferred")
print(f"{'D':>4s} {'rows':>12s} {'table':>9s} {'us':>8s} {'G reads/s':>10s} {'TB/s':>7s}")
for D in (16, 32, 64, 128):
    for rows in (1 << 20, 1 << 22, 1 << 25, 1 << 27, 1 << 29):
        if rows * D * 4 > 80e9:
            continue
        per = rows // NF
        base = [f * per for f in range(NF + 1)]
        table = torch.empty(per * NF, D, device=dev).normal_()
        g = torch.Generator(device=dev).manual_seed(rows + D)
        batches = [torch.randint(0, per, (B, NF), device=dev, generator=g) for _ in range(4)]
        ex_base = torch.tensor(base[:NF - 1], device=dev)

        def run(i):
            sp = batches[i % 4]
            return ops.fm2_fwd(table, None, sp[:, :NF - 1], ex_base, sp[:, NF - 1:], base[NF - 1], want_tsum=True)
        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(REPS):
            run(i)
        e.record()
        torch.cuda.synchronize()
        us = 1e3 * s.elapsed_time(e) / REPS
        n = B * NF
        print(f"{D:4d} {per * NF:12d} {per * NF * D * 4 / 2**30:8.2f}G {us:8.1f} {n / us / 1e3:10.1f} {n * D * 4 / us / 1e6:7.2f}", flush=True)
        del table, batches
        torch.cuda.empty_cache()
ops.raise_if_index_error()
