"""bf16 forward products of the tower shapes: rx_linear16_fwd (bias + ReLU fused, bf16 out) against the library's
bf16 product as torch reaches it (hipBLASLt; plain product, bf16 out), back to back with HIP events."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from recsys_amd import ops

dev = torch.device("cuda:0")
shapes = [(65536, 3344, 3344), (65536, 1672, 3344), (65536, 836, 1672), (65536, 1024, 432), (65536, 512, 1024), (65536, 256, 512),
          (65536, 432, 432), (65536, 216, 432), (65536, 108, 216)]
def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    x16 = ops.to_bf16(x)
    w16 = ops.to_bf16(w)
    xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
    t_mine = timed(lambda: ops.linear16_fwd(x16, w, b, True, True, w16=w16))
    t_lib = timed(lambda: torch.mm(xb, wb.t()))
    t_lib_epi = timed(lambda: torch._addmm_activation(b.to(torch.bfloat16), xb, wb.t()))
    fl = 2.0 * M * N * K
    print(f"M {M} N {N:5d} K {K:5d}: rx_linear16_fwd {t_mine:8.1f} us = {fl / t_mine / 1e6:7.1f} TF | library mm {t_lib:8.1f} us = {fl / t_lib / 1e6:7.1f} TF | "
          f"library addmm+relu {t_lib_epi:8.1f} us")
