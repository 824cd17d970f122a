#!/usr/bin/env python3
"""TFLOP/s of the dense-layer kernels on the tower shapes of the Criteo-sized configs.
Usage: python profiles/microbench_gemm.py [bf16=1] [iterations]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_amd import ops  # noqa: E402

bf16 = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
it = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e-3


shapes = [(65536, 1024, 432), (65536, 512, 1024), (65536, 256, 512), (65536, 432, 432)] if bf16 else \
         [(65536, 432, 432), (65536, 216, 432), (65536, 108, 216), (40960, 64, 128)]
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev)
    fl = 2.0 * M * N * K
    t1 = timeit(lambda: ops.linear_fwd(x, w, b, 1, bf16=bf16))
    t2 = timeit(lambda: ops.linear_bwd_x(dy, w, bf16=bf16))
    t3 = timeit(lambda: ops.linear_bwd_w(dy, x, bf16=bf16))
    print(f"M={M} N={N} K={K} {'bf16' if bf16 else 'f32'}: fwd {t1 * 1e6:7.1f} us {fl / t1 / 1e12:6.1f} TF | "
          f"bwd_x {t2 * 1e6:7.1f} us {fl / t2 / 1e12:6.1f} TF | bwd_w {t3 * 1e6:7.1f} us {fl / t3 / 1e12:6.1f} TF")
