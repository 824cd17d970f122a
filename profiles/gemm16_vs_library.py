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

print("weight gradients dW[N, K] = dY^T X (bias gradient included on this side), input gradients dX[M, K] = dY W")
for M, N, K in [(65536, 432, 432), (65536, 1024, 432), (65536, 512, 1024), (65536, 256, 512), (65536, 1672, 3344), (65536, 836, 1672)]:
    x16 = ops.to_bf16(torch.randn(M, K, device=dev))
    dy16 = ops.to_bf16(torch.randn(M, N, device=dev))
    w = torch.randn(N, K, device=dev) * 0.05
    xb, dyb, wb = x16.view(torch.bfloat16) if x16.dtype != torch.bfloat16 else x16, dy16, w.to(torch.bfloat16)
    t_w = timed(lambda: ops.linear16_bwd_w(dy16, x16, True))
    t_wl = timed(lambda: torch.mm(dyb.t(), xb))
    t_x = timed(lambda: ops.linear16_bwd_x(dy16, w, out_bf16=True))
    t_xl = timed(lambda: torch.mm(dyb, wb))
    fl = 2.0 * M * N * K
    print(f"M {M} N {N:5d} K {K:5d}: bwd_w {t_w:7.1f} us = {fl / t_w / 1e6:6.1f} TF | library {t_wl:7.1f} us = {fl / t_wl / 1e6:6.1f} TF || "
          f"bwd_x {t_x:7.1f} us = {fl / t_x / 1e6:6.1f} TF | library {t_xl:7.1f} us = {fl / t_xl / 1e6:6.1f} TF")
