#!/usr/bin/env python3
"""HIP-event timings of the dense-layer kernels at the NeuMF (MovieLens-1M, batch 8192 x 5 items)
shapes: M = 40960 rows, layers 128->64, 64->32, 64->32 (GMF), 64->1."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
M = 40960


def timeit(fn, it=50):
    """Per-call GPU time: `it` calls recorded into one HIP graph, so the host's launch cost stays out."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(it):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for N, K in ((64, 128), (32, 64), (1, 64)):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev)
    print(f"{K:4d}->{N:3d}  fwd {timeit(lambda: ops.linear_fwd(x, w, b, 1)):6.1f}  bwd_x {timeit(lambda: ops.linear_bwd_x(dy, w)):6.1f}"
          f"  bwd_w {timeit(lambda: ops.linear_bwd_w(dy, x)):6.1f} us")
