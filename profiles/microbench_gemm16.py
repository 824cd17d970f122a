#!/usr/bin/env python3
"""Back-to-back timing of the bf16-activation dense-layer ops at the DCN-v2 tower shapes (M = 65536),
next to torch's library bf16 matmul on the same shapes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recsys_amd import ops  # noqa: E402

M = int(os.environ.get("M", 65536))
dev = "cuda"


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for N, K in [(432, 432), (1024, 432), (512, 1024), (256, 512)]:
    x16 = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    w16 = ops.to_bf16(w)
    dy16 = torch.randn(M, N, device=dev).bfloat16()
    acc = torch.zeros(M, K, device=dev)
    fl = 2.0 * M * N * K
    f16 = t(lambda: ops.linear16_fwd(x16, w, b, relu=True, out_bf16=True, w16=w16))
    f32 = t(lambda: ops.linear16_fwd(x16, w, b, relu=False, out_bf16=False, w16=w16))
    bx = t(lambda: ops.linear16_bwd_x(dy16, w, gate16=x16, out_bf16=True))
    ba = t(lambda: ops.linear16_bwd_x(dy16, w, out=acc, accumulate=True))
    bw = t(lambda: ops.linear16_bwd_w(dy16, x16, True))
    lib = t(lambda: x16 @ w16.t())
    print(f"N={N:5d} K={K:5d}: fwd16 {f16:6.1f} ({fl / f16 / 1e6:4.0f} TF)  fwd32 {f32:6.1f}  bwdx-gated {bx:6.1f}  bwdx-acc {ba:6.1f}  "
          f"bwdw {bw:6.1f} ({fl / bw / 1e6:4.0f} TF)  lib fwd {lib:6.1f}", flush=True)
