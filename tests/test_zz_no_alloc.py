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


def test_noalloc_bce_recorded_in_a_graph_after_rx_init():
    """What the entry point is for: rx_init eagerly (set-up), then the call is RECORDED into a HIP graph
    and replayed twice with different inputs written into the same buffers, the second one shorter in
    effect (M is baked into the recording, so the tail is filled with a constant the reference op sees
    too).  Every replay must equal rx_bce_fwd on the same inputs bit for bit, and neither the recording
    nor the replays may take memory from the driver (free device memory unchanged) or from torch."""
    from recsys_amd import _lib
    lib = _lib.lib
    dev = torch.device("cuda", 0)
    _lib.check(lib.rx_init(0), "rx_init")
    _lib.check(lib.rx_init(0), "rx_init (idempotent)")
    M = 70_000
    logits = torch.zeros(M, device=dev)
    labels = torch.zeros(M, dtype=torch.float64, device=dev)
    loss_g = torch.zeros((), dtype=torch.float64, device=dev)
    dl_g = torch.zeros(M, device=dev)
    ws = torch.empty(int(lib.rx_bce_workspace_bytes(M)), dtype=torch.uint8, device=dev)
    loss_r = torch.zeros((), dtype=torch.float64, device=dev)
    dl_r = torch.zeros(M, device=dev)
    inputs = []
    for seed, fill_from in ((5, M), (6, 9_000)):
        g = torch.Generator(device=dev).manual_seed(seed)
        x = torch.randn(M, generator=g, device=dev) * 3
        y = (torch.rand(M, generator=g, device=dev) < 0.3).double()
        x[fill_from:] = 0.25
        y[fill_from:] = 1.0
        inputs.append((x, y))

    def ptr(t):
        return C.c_void_p(t.data_ptr())

    # an eager steady-state call first: no memory from torch or from the driver
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    stat0 = torch.cuda.memory_stats()["allocation.all.allocated"]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.rx_bce_fwd_noalloc(ptr(logits), ptr(labels), 1, M, ptr(loss_g), ptr(dl_g), 0, st), "rx_bce_fwd_noalloc")
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(dev)[0] == free0
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == stat0

    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.rx_bce_fwd_noalloc(ptr(logits), ptr(labels), 1, M, ptr(loss_g), ptr(dl_g), 0, st), "rx_bce_fwd_noalloc under capture")
    graph.replay()          # the first launch of an instantiated graph uploads it (the runtime's own memory)
    for x, y in inputs:
        logits.copy_(x); labels.copy_(y)
        loss_g.fill_(-1.0); dl_g.fill_(7.0)
        torch.cuda.synchronize()
        free1 = torch.cuda.mem_get_info(dev)[0]
        graph.replay()
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(dev)[0] == free1, "a replay took device memory from the driver"
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.rx_bce_fwd(ptr(logits), ptr(labels), 1, M, ptr(loss_r), ptr(dl_r), ptr(ws), ws.numel(), 0, st), "rx_bce_fwd")
        torch.cuda.synchronize()
        assert torch.equal(loss_g, loss_r), (loss_g.item(), loss_r.item())
        assert torch.equal(dl_g, dl_r)
        ref = torch.nn.functional.binary_cross_entropy_with_logits(x.double(), y)
        assert abs(loss_g.item() - ref.item()) <= 1e-12 * max(1.0, abs(ref.item()))


def test_noalloc_bce_first_call_under_capture_is_refused():
    """In a process where neither rx_init nor an eager call has run, a call on a capturing stream must
    fail with a clear message instead of resolving the device array's address inside the capture."""
    import os
    import subprocess
    import sys
    code = r"""
import ctypes as C, sys, torch
sys.path.insert(0, %r)
from recsys_amd import _lib
lib = _lib.lib
x = torch.zeros(256, device="cuda"); y = torch.zeros(256, dtype=torch.float64, device="cuda")
loss = torch.zeros((), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
rc = None
with torch.cuda.stream(s):
    g.capture_begin()
    rc = lib.rx_bce_fwd_noalloc(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), 1, 256, C.c_void_p(loss.data_ptr()), None, 0,
                                C.c_void_p(s.cuda_stream))
    msg = lib.rx_last_error()
    g.capture_end()
assert rc != 0 and b"rx_init" in msg, (rc, msg)
_lib.check(lib.rx_init(0), "rx_init")
rc = lib.rx_bce_fwd_noalloc(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), 1, 256, C.c_void_p(loss.data_ptr()), None, 0,
                            C.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
assert rc == 0 and abs(loss.item() - 0.6931471805599453) < 1e-15
print("refused-then-ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "refused-then-ok" in p.stdout, p.stderr[-2000:]


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
