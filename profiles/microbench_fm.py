#!/usr/bin/env python3
"""Runs rx_segreduce_fm (COMPACT + sum of squares) a few times on Criteo-shaped keys; meant to sit
under rocprofv3 --pmc.  Usage: python profiles/microbench_fm.py [iterations]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_amd import _lib, ops, synthetic  # noqa: E402

it = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dim, B = 16, 65536
dev = torch.device("cuda", 0)
dims = synthetic.criteo_field_dims()
b = synthetic.criteo_batch(dims, B, 0, dev)
offs = torch.tensor([0] + dims[:-1], device=dev).cumsum(0)
rows = sum(dims)
F = len(dims) - 1
flat_fm = torch.cat([(b["sparse"][:, :F] + offs[None, :F]).reshape(-1), b["sparse"][:, F] + offs[F]]).contiguous()
si = ops.sort_dedup(flat_fm, rows)
table = torch.randn(rows, dim, device=dev)
T = torch.randn(B, dim, device=dev)
gl = torch.randn(B, device=dev) / B
compact = torch.empty(flat_fm.numel(), dim, device=dev)
compact2 = torch.empty(flat_fm.numel(), device=dev)
for _ in range(it):
    ops.segreduce_fm(T, gl, F, si, _lib.RX_SEG_COMPACT, compact, table=table, target2=compact2, want_sq=True)
torch.cuda.synchronize()
