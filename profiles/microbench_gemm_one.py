#!/usr/bin/env python3
"""A few launches of one dense-layer kernel (meant to sit under rocprofv3 --pmc).
Usage: python profiles/microbench_gemm_one.py fwd|bwd_x|bwd_w [bf16=1] [M N K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
bf16 = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
M, N, K = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (65536, 1024, 432)
dev = torch.device("cuda", 0)
x = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev) / K ** 0.5
b = torch.randn(N, device=dev)
dy = torch.randn(M, N, device=dev)
for _ in range(5):
    if which == "fwd":
        ops.linear_fwd(x, w, b, 1, bf16=bf16)
    elif which == "bwd_x":
        ops.linear_bwd_x(dy, w, bf16=bf16)
    else:
        ops.linear_bwd_w(dy, x, bf16=bf16)
torch.cuda.synchronize()
