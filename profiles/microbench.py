#!/usr/bin/env python3
"""Micro-timings (HIP events) of the index sort and the segmented reduce on Criteo-shaped keys.
Usage: python profiles/microbench.py [dim] [mode]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_amd import _lib, ops, synthetic  # noqa: E402

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
dims = synthetic.criteo_field_dims()
b = synthetic.criteo_batch(dims, 65536, 0, dev)
offs = torch.tensor([0] + dims[:-1], device=dev).cumsum(0)
flat = (b["sparse"] + offs[None, :]).reshape(-1).contiguous()
n, rows = flat.numel(), sum(dims)
occ = torch.randn(n, dim, device=dev)
occ2 = torch.randn(n, device=dev)
table = torch.randn(rows, dim, device=dev)
table2 = torch.randn(rows, 1, device=dev)


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


si = ops.sort_dedup(flat, rows)
nu = int(si.n_uniq.item())
print(f"n={n} rows={rows} unique={nu} rho={nu / n:.3f} dim={dim}")
print(f"sort_dedup            {timeit(lambda: ops.sort_dedup(flat, rows)):8.1f} us")
compact = torch.empty(n, dim, device=dev)
compact2 = torch.empty(n, device=dev)
print(f"segreduce COMPACT     {timeit(lambda: ops.segreduce(occ, si, _lib.RX_SEG_COMPACT, compact)):8.1f} us")
print(f"segreduce COMPACT+2+sq{timeit(lambda: ops.segreduce(occ, si, _lib.RX_SEG_COMPACT, compact, occ2=occ2, target2=compact2, want_sq=True)):8.1f} us")
print(f"segreduce SGD         {timeit(lambda: ops.segreduce(occ, si, _lib.RX_SEG_SGD, table, lr=1e-3)):8.1f} us")
print(f"segreduce SGD+2       {timeit(lambda: ops.segreduce(occ, si, _lib.RX_SEG_SGD, table, lr=1e-3, occ2=occ2, target2=table2)):8.1f} us")
st = torch.zeros(rows, dim, device=dev)
print(f"segreduce ADAGRAD     {timeit(lambda: ops.segreduce(occ, si, _lib.RX_SEG_ADAGRAD, table, state=st, lr=1e-3, eps=1e-10)):8.1f} us")
print(f"row_update SGD        {timeit(lambda: ops.row_update(compact, si, _lib.RX_SEG_SGD, table, lr=1e-3)):8.1f} us")
F = len(dims) - 1
B = 65536
flat_fm = torch.cat([(b["sparse"][:, :F] + offs[None, :F]).reshape(-1), b["sparse"][:, F] + offs[F]]).contiguous()
si_fm = ops.sort_dedup(flat_fm, rows)
ex, item = b["sparse"][:, :F], b["sparse"][:, F:]
ex_base, item_base = offs[:F].contiguous(), int(offs[F])
print(f"flat list + sort_dedup {timeit(lambda: ops.sort_dedup(ops.fm2_flat_indices(table, ex, ex_base, item, item_base), rows)):8.1f} us")
print(f"fm_sort_dedup (no flat list) {timeit(lambda: ops.fm_sort_dedup(table, ex, ex_base, item, item_base)):8.1f} us")
T = torch.randn(B, dim, device=dev)
gl = torch.randn(B, device=dev) / B
print(f"segreduce_fm COMPACT+sq{timeit(lambda: ops.segreduce_fm(T, gl, F, si_fm, _lib.RX_SEG_COMPACT, compact, table=table, target2=compact2, want_sq=True)):8.1f} us")
print(f"segreduce_fm SGD      {timeit(lambda: ops.segreduce_fm(T, gl, F, si_fm, _lib.RX_SEG_SGD, table, target2=table2, lr=1e-3)):8.1f} us")
x = torch.empty(n * dim, device=dev)
print(f"copy {n * dim * 4 / 1e6:.0f} MB (reference) {timeit(lambda: x.copy_(occ.view(-1))):8.1f} us")
