"""The entry point that allocates nothing (rx_bce_fwd_noalloc): called straight through the C ABI
with caller-owned inputs and outputs only, checked against rx_bce_fwd bit for bit on two calls in a
row, the second with a smaller input (the library's device array keeps the first call's contents)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(M, seed, label_dtype):
    from recsys_amd import _lib
    lib = _lib.lib
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(seed)
    logits = (torch.randn(M, generator=g, device=dev) * 3).contiguous()
    labels = (torch.rand(M, generator=g, device=dev) < 0.3).to(label_dtype).contiguous()
    f64 = int(label_dtype == torch.float64)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def ptr(t):
        return C.c_void_p(t.data_ptr())

    # existing entry point, caller workspace
    ws = torch.empty(int(lib.rx_bce_workspace_bytes(M)), dtype=torch.uint8, device=dev)
    loss_a = torch.full((), -1.0, dtype=torch.float64, device=dev)
    dl_a = torch.full((M,), 7.0, device=dev)
    _lib.check(lib.rx_bce_fwd(ptr(logits), ptr(labels), f64, M, ptr(loss_a), ptr(dl_a), ptr(ws), ws.numel(), 0, stream), "rx_bce_fwd")
    # the variant without a workspace: every buffer it gets exists before the call
    loss_b = torch.full((), -2.0, dtype=torch.float64, device=dev)
    dl_b = torch.full((M,), 9.0, device=dev)
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats()["allocation.all.allocated"]
    _lib.check(lib.rx_bce_fwd_noalloc(ptr(logits), ptr(labels), f64, M, ptr(loss_b), ptr(dl_b), 0, stream), "rx_bce_fwd_noalloc")
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == before
    return (loss_a, dl_a), (loss_b, dl_b), (logits, labels)


def test_noalloc_bce_matches_workspace_variant_twice_in_a_row():
    # 300000 logits fill all 1024 partial slots; the second, smaller call uses 40 of them and must not
    # see what the first one left in the rest
    for M, seed, dt in ((300_000, 1, torch.float64), (10_001, 2, torch.float64), (513, 3, torch.float32)):
        (la, da), (lb, db), (x, y) = _both(M, seed, dt)
        assert torch.equal(la, lb), (M, la.item(), lb.item())
        assert torch.equal(da, db), M
        ref = torch.nn.functional.binary_cross_entropy_with_logits(x.double(), y.double())
        assert abs(lb.item() - ref.item()) <= 1e-12 * max(1.0, abs(ref.item()))


def test_noalloc_bce_rejects_oversize_and_null():
    from recsys_amd import _lib
    lib = _lib.lib
    x = torch.zeros(4, device="cuda")
    y = torch.zeros(4, dtype=torch.float64, device="cuda")
    loss = torch.zeros((), dtype=torch.float64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.rx_bce_fwd_noalloc(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), 1, (1 << 28) + 1, C.c_void_p(loss.data_ptr()), None, 0, st)
    assert rc != 0 and b"beyond the supported maximum" in lib.rx_last_error()
    rc = lib.rx_bce_fwd_noalloc(None, C.c_void_p(y.data_ptr()), 1, 4, C.c_void_p(loss.data_ptr()), None, 0, st)
    assert rc != 0 and b"null" in lib.rx_last_error()
