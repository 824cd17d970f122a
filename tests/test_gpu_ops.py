"""Kernel-level parity on the GPU, called through the C ABI (recsys_amd.ops wraps ctypes 1:1).

Integer / index results (gathered rows, sort order, distinct-row sets, run offsets, owner
buckets, top-k codes) must be bit-exact against torch / numpy on the CPU; floating-point results
are compared with rtol 1e-5 against the fp32 CPU oracle (tolerance scaled by the tensor's
magnitude, see conftest.assert_close)."""
import numpy as np
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from recsys_amd import ops as o
    return o


def _gen(seed):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------ gather
@pytest.mark.parametrize("dim", [1, 3, 16, 20, 64, 128, 260])
def test_gather_rows_bit_exact(ops, dim):
    g = _gen(dim)
    table = torch.randn(97, dim, generator=g)
    users = torch.randint(0, 97, (33,), generator=g)
    items = torch.randint(0, 97, (33, 5), generator=g)
    ex = torch.randint(0, 30, (33, 3), generator=g)
    td, ud, idd, exd = table.cuda(), users.cuda(), items.cuda(), ex.cuda()
    segs = [ops.Segment(td, ud, 0, div=5), ops.Segment(td, idd, dim), ops.Segment(td, exd, 2 * dim, div=5, stride=3, k=1, base=7)]
    out = ops.gather_concat(segs, 33 * 5, 3 * dim).cpu()
    exp = torch.cat([table[users][:, None, :].expand(33, 5, dim), table[items], table[ex[:, 1] + 7][:, None, :].expand(33, 5, dim)], -1)
    assert torch.equal(out, exp.reshape(33 * 5, 3 * dim))


@pytest.mark.parametrize("dims,rows,N", [((64, 64), 40960, 5), ((16, 32, 8, 128), 1000, 3), ((16, 7), 300, 2), ((64,) * 8, 129, 1)])
def test_gather_many_segments_in_one_launch(ops, dims, rows, N):
    """Segments of different widths, tables and index layouts (broadcast, field pick) at sizes that
    span many 64-row tiles with a ragged last one; (16, 7) forces the per-segment fallback."""
    g = _gen(rows)
    B = rows // N if rows % N == 0 else rows
    N = rows // B
    fields = torch.randint(0, 50, (B, len(dims)), generator=g)
    items = torch.randint(0, 80, (B, N), generator=g)
    tables = [torch.randn(90, d, generator=g) for d in dims]
    segs, exp, col = [], [], 0
    for q, (t, d) in enumerate(zip(tables, dims)):
        if q % 2 == 0:
            segs.append(ops.Segment(t.cuda(), items.cuda(), col))
            exp.append(t[items].reshape(B * N, d))
        else:
            segs.append(ops.Segment(t.cuda(), fields.cuda(), col, div=N, stride=len(dims), k=q, base=3))
            exp.append(t[fields[:, q] + 3][:, None, :].expand(B, N, d).reshape(B * N, d))
        col += d
    pad = (-col) % 4 if all(d % 4 == 0 for d in dims) else 0
    out = ops.gather_concat(segs, B * N, col + pad)
    assert torch.equal(out[:, :col].cpu(), torch.cat(exp, 1))


@pytest.mark.parametrize("dim,bag", [(16, 3), (64, 8), (5, 4), (128, 26)])
def test_gather_sum_pool(ops, dim, bag):
    """bag > 1: a block is the sum, in bag order, of a bag of rows -- against an fp32 loop in the same
    order (bit for bit), forward; backward through autograd against torch's index + sum."""
    from recsys_amd import autograd as rxa
    g = _gen(dim * bag)
    B = 257
    table = torch.randn(400, dim, generator=g)
    other = torch.randn(50, dim, generator=g)
    idx = torch.randint(0, 300, (B, bag + 2), generator=g)
    idx[:, 0] %= 50
    segs = [ops.Segment(table.cuda(), idx.cuda(), 0, stride=bag + 2, k=1, base=7, bag=bag),
            ops.Segment(other.cuda(), idx.cuda(), dim, stride=bag + 2, k=0, base=0)]
    out = ops.gather_concat(segs, B, 2 * dim).cpu()
    acc = torch.zeros(B, dim)
    for m in range(bag):
        acc = acc + table[idx[:, 1 + m] + 7]
    assert torch.equal(out[:, :dim], acc)
    assert torch.equal(out[:, dim:], other[idx[:, 0]])
    # autograd: pooled block + plain block
    wt = table.cuda().requires_grad_(True)
    spec = [(0, idx.cuda(), 0, 1, bag + 2, 1, 7, bag)]
    y = rxa.GatherConcat.apply(spec, B, dim, wt)
    w = torch.randn(B, dim, generator=g)
    (y * w.cuda()).sum().backward()
    ref = table.clone().requires_grad_(True)
    (ref[idx[:, 1:1 + bag] + 7].sum(1) * w).sum().backward()
    assert_close(wt.grad.cpu().numpy(), ref.grad.numpy(), rtol=1e-5, what="pooled gather grad")


def test_gather_out_of_range_raises(ops):
    table = torch.randn(10, 8).cuda()
    bad = torch.tensor([0, 10, 3]).cuda()
    with pytest.raises(IndexError):
        ops.gather_concat([ops.Segment(table, bad, 0)], 3, 8)
    neg = torch.tensor([0, -1, 3]).cuda()
    with pytest.raises(IndexError):
        ops.gather_concat([ops.Segment(table, neg, 0)], 3, 8)
    ok = torch.tensor([0, 9, 3]).cuda()
    ops.gather_concat([ops.Segment(table, ok, 0)], 3, 8)


def test_empty_batch(ops):
    table = torch.randn(10, 8).cuda()
    out = ops.gather_concat([ops.Segment(table, torch.zeros(0, dtype=torch.int64).cuda(), 0)], 0, 8)
    assert out.shape == (0, 8)
    si = ops.sort_dedup(torch.zeros(0, dtype=torch.int64).cuda(), 10)
    assert int(si.n_uniq.item()) == 0


# ------------------------------------------------------------------ sort / dedup
@pytest.mark.parametrize("n,n_rows", [(1, 1), (5, 3), (64, 7), (1000, 50), (4096, 4096), (4097, 100000),
                                      (70001, 6040), (300000, 33_000_000), (200000, 2**32),
                                      (2047, 9), (2048, 2000), (2049, 5), (2_100_003, 33_000_000),
                                      # around the limit of the one-launch path (8192 positions), with one,
                                      # two, three and four 8-bit passes
                                      (63, 200), (65, 256), (511, 257), (513, 70000), (5120, 3706), (8191, 6040),
                                      (8192, 2**24), (8192, 2**32), (8193, 6040), (7000, 2**32),
                                      # small tables: one launch, a workgroup per range of 64 rows (up to 16384 rows
                                      # and 4 Mi positions read over all workgroups), and just past its limits
                                      (8192, 6040), (49152, 3706), (16, 16384), (255, 16384), (256, 16385), (65536, 4096),
                                      (65537, 4096), (1025, 64), (1023, 65), (100000, 63)])
def test_sort_dedup_bit_exact(ops, n, n_rows):
    g = _gen(n)
    if n_rows > 2**31:
        idx = torch.randint(0, 2**31, (n,), generator=g) * 2 + torch.randint(0, 2, (n,), generator=g)
    else:
        # skewed like item popularity: many repeats of small ids plus a uniform tail
        idx = torch.where(torch.rand(n, generator=g) < 0.5, torch.randint(0, max(n_rows // 50, 1), (n,), generator=g),
                          torch.randint(0, n_rows, (n,), generator=g))
    si = ops.sort_dedup(idx.cuda(), n_rows, want_inverse=True)
    nu = int(si.n_uniq.item())
    exp_sorted, exp_pos = torch.sort(idx, stable=True)
    uniq, inv, counts = torch.unique(idx, sorted=True, return_inverse=True, return_counts=True)
    assert nu == uniq.numel()
    assert torch.equal(si.sorted_pos.cpu().long(), exp_pos)
    assert torch.equal(si.uniq_rows[:nu].cpu(), uniq)
    seg = si.seg_off[: nu + 1].cpu().long()
    assert torch.equal(seg[1:] - seg[:-1], counts)
    assert int(seg[0]) == 0 and int(seg[-1]) == n
    assert torch.equal(si.inverse.cpu().long(), inv)
    assert torch.equal(si.sorted_slot.cpu().long(), inv[exp_pos])


@pytest.mark.parametrize("B,F,N,pad", [(1, 1, 1, 0), (77, 3, 1, 2), (5000, 25, 1, 0), (3001, 7, 4, 1)])
def test_fm_sort_dedup_equals_sort_of_flat_list(ops, B, F, N, pad):
    # keys read straight from the index matrices (strided views included) == sorting the flat list
    g = _gen(B + F)
    dims = [int(x) for x in torch.randint(1, 70000, (F + 1,), generator=g)]
    base = torch.tensor([sum(dims[:f]) for f in range(F)]).cuda()
    item_base, n_feat = sum(dims[:F]), sum(dims)
    both = torch.stack([torch.randint(0, d, (B,), generator=g) for d in dims[:F]] +
                       [torch.randint(0, dims[F], (B,), generator=g) for _ in range(N + pad)], 1).cuda()
    ex, item = both[:, :F], both[:, F:F + N]
    feat = torch.zeros(n_feat, 4).cuda()
    a = ops.fm_sort_dedup(feat, ex, base, item, item_base)
    b = ops.sort_dedup(ops.fm2_flat_indices(feat, ex, base, item, item_base), n_feat)
    nu = int(a.n_uniq.item())
    assert nu == int(b.n_uniq.item())
    for x, y in ((a.sorted_pos, b.sorted_pos), (a.sorted_slot, b.sorted_slot), (a.uniq_rows[:nu], b.uniq_rows[:nu]),
                 (a.seg_off[: nu + 1], b.seg_off[: nu + 1])):
        assert torch.equal(x, y)
    ops.raise_if_index_error()


def test_sort_dedup_is_deterministic(ops):
    idx = torch.randint(0, 1000, (50000,), generator=_gen(1)).cuda()
    a = ops.sort_dedup(idx, 1000)
    b = ops.sort_dedup(idx, 1000)
    nu = int(a.n_uniq.item())
    assert nu == int(b.n_uniq.item())
    assert torch.equal(a.sorted_pos, b.sorted_pos) and torch.equal(a.seg_off[: nu + 1], b.seg_off[: nu + 1])
    assert torch.equal(a.uniq_rows[:nu], b.uniq_rows[:nu])


def test_sort_dedup_flags_out_of_range(ops):
    idx = torch.tensor([1, 5, 2]).cuda()
    ops.sort_dedup(idx, 5)
    with pytest.raises(IndexError):
        ops.raise_if_index_error()


# ------------------------------------------------------------------ segmented reduce + updates
@pytest.mark.parametrize("dim", [1, 5, 16, 64, 128])
def test_segreduce_dense_matches_index_add(ops, dim):
    from recsys_amd import _lib
    g = _gen(dim + 3)
    n, rows = 5000, 300
    idx = torch.randint(0, rows, (n,), generator=g)
    occ = torch.randn(n, dim, generator=g)
    si = ops.sort_dedup(idx.cuda(), rows)
    grad = torch.zeros(rows, dim).cuda()
    sq = ops.segreduce(occ.cuda(), si, _lib.RX_SEG_DENSE, grad, want_sq=True)
    exp = torch.zeros(rows, dim, dtype=torch.float64).index_add_(0, idx, occ.double())
    assert_close(grad.cpu().numpy(), exp.numpy(), what="dense grad")
    assert_close(float(sq.double().sum().item()), float((exp ** 2).sum()), what="sum of squares")
    # run-to-run bitwise reproducibility
    grad2 = torch.zeros(rows, dim).cuda()
    ops.segreduce(occ.cuda(), si, _lib.RX_SEG_DENSE, grad2)
    assert torch.equal(grad, grad2)
    # untouched rows are exactly zero, touched-row set is exact
    touched = torch.zeros(rows, dtype=torch.bool)
    touched[idx] = True
    assert torch.equal((grad.cpu() != 0).any(1) | ~touched, torch.ones(rows, dtype=torch.bool)) or dim == 1
    assert not grad.cpu()[~touched].any()


@pytest.mark.parametrize("mode", ["sgd", "adagrad", "rowwise_adagrad"])
@pytest.mark.parametrize("dim", [1, 16, 64])
def test_fused_row_update_matches_torch_optim(ops, mode, dim):
    from recsys_amd import _lib
    g = _gen(11)
    n, rows, lr = 3000, 200, 0.1
    w0 = torch.randn(rows, dim, generator=g)
    w_ref = w0.clone().requires_grad_(True)
    w_dev = w0.clone().cuda()
    state = None
    if mode == "adagrad":
        state = torch.zeros(rows, dim).cuda()
        opt = torch.optim.Adagrad([w_ref], lr=lr)
    elif mode == "rowwise_adagrad":
        state = torch.zeros(rows).cuda()
        opt = None
        ref_state = torch.zeros(rows)
    else:
        opt = torch.optim.SGD([w_ref], lr=lr)
    code = {"sgd": _lib.RX_SEG_SGD, "adagrad": _lib.RX_SEG_ADAGRAD, "rowwise_adagrad": _lib.RX_SEG_ROWWISE_ADAGRAD}[mode]
    for step in range(3):
        idx = torch.randint(0, rows // 2, (n,), generator=g)  # upper half never touched
        occ = torch.randn(n, dim, generator=g)
        si = ops.sort_dedup(idx.cuda(), rows)
        ops.segreduce(occ.cuda(), si, code, w_dev, state=state, lr=lr, eps=1e-10)
        grad = torch.zeros(rows, dim).index_add_(0, idx, occ)
        if opt is not None:
            opt.zero_grad()
            w_ref.grad = grad
            opt.step()
        else:
            with torch.no_grad():
                ref_state += (grad ** 2).mean(1)
                touched = torch.zeros(rows, dtype=torch.bool)
                touched[idx] = True
                upd = lr * grad / (ref_state.sqrt()[:, None] + 1e-10)
                w_ref -= torch.where(touched[:, None], upd, torch.zeros_like(upd))
    assert_close(w_dev.cpu().numpy(), w_ref.detach().numpy(), what=f"{mode} weights")
    assert torch.equal(w_dev.cpu()[rows // 2:], w0[rows // 2:]), "untouched rows must not move"


def test_two_phase_clip_update(ops):
    from recsys_amd import _lib
    g = _gen(5)
    n, rows, dim, lr = 2000, 100, 16, 0.5
    w0 = torch.randn(rows, dim, generator=g)
    idx = torch.randint(0, rows, (n,), generator=g)
    occ = torch.randn(n, dim, generator=g)
    si = ops.sort_dedup(idx.cuda(), rows)
    compact = torch.empty(n, dim).cuda()
    sq = ops.segreduce(occ.cuda(), si, _lib.RX_SEG_COMPACT, compact, want_sq=True)
    acc = torch.zeros(1, dtype=torch.float64).cuda()
    ops.sum_accumulate(sq, acc)
    scale, total = ops.clip_scale(acc, 1.0)
    w = w0.clone().cuda()
    ops.row_update(compact, si, _lib.RX_SEG_SGD, w, lr=lr, scale=scale)
    grad = torch.zeros(rows, dim).index_add_(0, idx, occ)
    wr = w0.clone().requires_grad_(True)
    wr.grad = grad.clone()
    tn = torch.nn.utils.clip_grad_norm_([wr], 1.0)
    assert_close(total.item(), tn.item(), what="total norm")
    with torch.no_grad():
        wr -= lr * wr.grad
    assert_close(w.cpu().numpy(), wr.detach().numpy(), what="clipped sgd")


@pytest.mark.parametrize("mode", ["sgd", "adagrad"])
def test_two_phase_clip_update_with_companion(ops, mode):
    """Compact gradients of a table and of its [rows, 1] companion, clip factor formed in one launch
    from the per-block partials plus a dense-gradient accumulator, both tables updated by one launch."""
    from recsys_amd import _lib
    g = _gen(6)
    n, rows, dim, lr = 3000, 150, 16, 0.5
    w0, b0 = torch.randn(rows, dim, generator=g), torch.randn(rows, 1, generator=g)
    idx = torch.randint(0, rows, (n,), generator=g)
    occ, occ2 = torch.randn(n, dim, generator=g), torch.randn(n, generator=g)
    dense_g = torch.randn(37, generator=g)
    si = ops.sort_dedup(idx.cuda(), rows)
    compact, compact2 = torch.empty(n, dim).cuda(), torch.empty(n).cuda()
    sq = ops.segreduce(occ.cuda(), si, _lib.RX_SEG_COMPACT, compact, want_sq=True, occ2=occ2.cuda(), target2=compact2)
    acc = torch.zeros(1, dtype=torch.float64).cuda()
    ops.sumsq_accumulate(dense_g.cuda(), acc)
    scale, total = ops.clip_scale_parts(sq, acc, 1.0)
    w, b = w0.clone().cuda(), b0.clone().cuda()
    kind = {"sgd": _lib.RX_SEG_SGD, "adagrad": _lib.RX_SEG_ADAGRAD}[mode]
    st = torch.zeros(rows, dim).cuda() if mode == "adagrad" else None
    st2 = torch.zeros(rows, 1).cuda() if mode == "adagrad" else None
    ops.row_update(compact, si, kind, w, state=st, lr=lr, eps=1e-10, scale=scale, compact2=compact2, table2=b, state2=st2)
    wr, br = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    dr = torch.zeros(37, requires_grad=True)
    wr.grad = torch.zeros(rows, dim).index_add_(0, idx, occ)
    br.grad = torch.zeros(rows, 1).index_add_(0, idx, occ2[:, None])
    dr.grad = dense_g.clone()
    tn = torch.nn.utils.clip_grad_norm_([wr, br, dr], 1.0)
    assert_close(total.item(), tn.item(), what="total norm")
    opt = torch.optim.SGD([wr, br], lr=lr) if mode == "sgd" else torch.optim.Adagrad([wr, br], lr=lr, eps=1e-10)
    opt.step()
    touched = torch.zeros(rows, dtype=torch.bool)
    touched[idx] = True
    assert_close(w.cpu().numpy()[touched], wr.detach().numpy()[touched], what=f"{mode} table")
    assert_close(b.cpu().numpy()[touched], br.detach().numpy()[touched], what=f"{mode} companion")
    assert torch.equal(w.cpu()[~touched], w0[~touched]) and torch.equal(b.cpu()[~touched], b0[~touched])


@pytest.mark.parametrize("mode", ["sgd", "adagrad"])
def test_multi_tensor_norm_and_update(ops, mode):
    """Sum of squares and optimizer step of many small dense tensors, one launch each."""
    from recsys_amd import _lib
    g = _gen(11)
    shapes = [(64, 128), (64,), (1, 32), (1,), (300, 7), (5000,)]
    ws = [torch.randn(*sh, generator=g) for sh in shapes]
    gs = [torch.randn(*sh, generator=g) for sh in shapes]
    wd, gd = [w.clone().cuda() for w in ws], [x.clone().cuda() for x in gs]
    parts64 = ops.multi_sumsq_partials([x.reshape(-1) for x in gd])
    scale, total = ops.clip_scale_parts(None, None, 1.0, parts64)
    ref = [w.clone().requires_grad_(True) for w in ws]
    for r, x in zip(ref, gs):
        r.grad = x.clone()
    tn = torch.nn.utils.clip_grad_norm_(ref, 1.0)
    assert_close(total.item(), tn.item(), what="total norm")
    opt = torch.optim.SGD(ref, lr=0.3) if mode == "sgd" else torch.optim.Adagrad(ref, lr=0.3, eps=1e-10)
    opt.step()
    states = [torch.zeros_like(w) for w in wd] if mode == "adagrad" else None
    kind = _lib.RX_SEG_SGD if mode == "sgd" else _lib.RX_SEG_ADAGRAD
    ops.multi_dense_update(wd, gd, kind, 0.3, 1e-10, scale=scale, states=states)
    for a, b in zip(wd, ref):
        assert_close(a.cpu().numpy(), b.detach().numpy(), what=f"{mode} dense step")


# ------------------------------------------------------------------ BatchNorm1d + ReLU + dropout
@pytest.mark.parametrize("M,C", [(300, 70), (1024, 64), (4096, 216), (2, 5)])
@pytest.mark.parametrize("relu,drop", [(False, False), (True, False), (True, True)])
def test_batch_norm_act_matches_torch(ops, M, C, relu, drop):
    g = _gen(M + C)
    x = torch.randn(M, C, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    keep = (torch.rand(M, C, generator=g) < 0.9).to(torch.uint8) if drop else None
    scale = 1.0 / 0.9 if drop else 1.0
    dout = torch.randn(M, C, generator=g)
    eps, mom = 1e-5, 0.1
    # torch reference in fp64
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    y = torch.nn.functional.batch_norm(xr, rm_ref, rv_ref, gr, br, training=True, momentum=mom, eps=eps)
    if relu:
        y = y.relu()
    if drop:
        y = y * keep.double() * scale
    y.backward(dout.double())
    # device
    xd = x.cuda()
    rmd, rvd = rm.clone().cuda(), rv.clone().cuda()
    mean, rstd = ops.bn_stats(xd, eps, mom, rmd, rvd)
    out = ops.bn_apply(xd, mean, rstd, gamma.cuda(), beta.cuda(), relu, keep.cuda() if drop else None, scale)
    assert_close(out.cpu().numpy(), y.detach().numpy(), what="bn out")
    assert_close(rmd.cpu().numpy(), rm_ref.numpy(), what="running mean")
    assert_close(rvd.cpu().numpy(), rv_ref.numpy(), what="running var")
    dx, dgamma, dbeta = ops.bn_bwd(dout.cuda(), xd, out, mean, rstd, gamma.cuda(), relu or drop, scale, True)
    # dx is a difference of O(1) terms; with two rows it cancels to ~1e-4, so allow fp32 rounding of the terms
    assert_close(dx.cpu().numpy(), xr.grad.numpy(), what="bn dx", atol=5e-6)
    assert_close(dgamma.cpu().numpy(), gr.grad.numpy(), what="bn dgamma")
    assert_close(dbeta.cpu().numpy(), br.grad.numpy(), what="bn dbeta")


def test_batch_norm_act_module_eval_and_train(ops):
    """The tower helper against torch modules: training statistics / buffers, then evaluation mode."""
    from recsys_amd import autograd as rxa
    torch.manual_seed(3)
    bn = torch.nn.BatchNorm1d(48).cuda()
    ref = torch.nn.BatchNorm1d(48).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_()
    ref.load_state_dict(bn.state_dict())
    x = torch.randn(512, 48, device="cuda")
    for _ in range(2):
        a = rxa.batch_norm_act(x, bn, True, 0.0)
        b = ref(x).relu()
        assert_close(a.detach().cpu().numpy(), b.detach().cpu().numpy(), what="train fwd")
    assert int(bn.num_batches_tracked) == 2
    assert_close(bn.running_var.cpu().numpy(), ref.running_var.cpu().numpy(), what="running var")
    bn.eval(); ref.eval()
    assert_close(rxa.batch_norm_act(x, bn, False, 0.3).detach().cpu().numpy(), ref(x).detach().cpu().numpy(), what="eval fwd")


# ------------------------------------------------------------------ dense layers
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (7, 3, 5), (130, 32, 64), (257, 64, 128), (1000, 1, 32),
                                   (513, 107, 214), (300, 214, 429), (2048, 128, 256)])
@pytest.mark.parametrize("act", [0, 1])
def test_linear_fp32_matches_cpu(ops, M, N, K, act):
    g = _gen(M * 7 + N)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    y = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda(), act).cpu()
    exp = torch.nn.functional.linear(x.double(), w.double(), b.double())
    if act:
        exp = exp.relu()
    assert_close(y.numpy(), exp.numpy(), what="linear fwd")
    dy = torch.randn(M, N, generator=g)
    dx = ops.linear_bwd_x(dy.cuda(), w.cuda()).cpu()
    assert_close(dx.numpy(), (dy.double() @ w.double()).numpy(), what="linear bwd x")
    dw, db = ops.linear_bwd_w(dy.cuda(), x.cuda())
    assert_close(dw.cpu().numpy(), (dy.double().T @ x.double()).numpy(), what="linear bwd w")
    assert_close(db.cpu().numpy(), dy.double().sum(0).numpy(), what="linear bwd b")


@pytest.mark.parametrize("M,K", [(1, 4), (37, 64), (1000, 108), (4099, 256), (40960, 64), (65536, 108), (300, 20)])
def test_linear_single_output_column(ops, M, K):
    """Linear(K, 1), the last layer of every tower, takes the matrix-vector kernels: forward with and
    without ReLU / bias, grad-X plain and accumulating, grad-W + grad-b, on a column block of a wider
    matrix as well (row stride > K)."""
    g = _gen(M + K)
    wide = torch.randn(M, K + 8, generator=g).cuda()
    x = wide[:, 4:4 + K]                      # 16-byte aligned rows with a row stride of K + 8
    w, b = (torch.randn(1, K, generator=g) / K ** 0.5).cuda(), torch.randn(1, generator=g).cuda()
    xd, wd = x.double().cpu(), w.double().cpu()
    for act in (0, 1):
        y = ops.linear_fwd(x, w, b, act).cpu()
        exp = xd @ wd.T + b.double().cpu()
        assert_close(y.numpy(), (exp.relu() if act else exp).numpy(), what="fwd")
    assert_close(ops.linear_fwd(x, w, None, 0).cpu().numpy(), (xd @ wd.T).numpy(), what="fwd without bias")
    dy = torch.randn(M, 1, generator=g).cuda()
    dx = ops.linear_bwd_x(dy, w)
    assert_close(dx.cpu().numpy(), (dy.double().cpu() @ wd).numpy(), what="bwd x")
    acc = torch.randn(M, K, generator=g).cuda()
    exp_acc = acc.double().cpu() + dy.double().cpu() @ wd
    ops.linear_bwd_x(dy, w, out=acc, accumulate=True)
    assert_close(acc.cpu().numpy(), exp_acc.numpy(), what="bwd x accumulating")
    dw, db = ops.linear_bwd_w(dy, x)
    assert_close(dw.cpu().numpy(), (dy.double().cpu().T @ xd).numpy(), what="bwd w")
    assert_close(db.cpu().numpy(), dy.double().cpu().sum(0).numpy(), what="bwd b")
    dw_only, none = ops.linear_bwd_w(dy, x, want_bias=False)
    assert none is None and torch.equal(dw_only, dw)
    dw2, db2 = ops.linear_bwd_w(dy, x)
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "run-to-run bits"


def test_linear_tile_layout_asymmetric(ops):
    """A = I against an asymmetric B catches a transposed accumulator write."""
    n = 96
    x = torch.eye(n)
    w = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 97) - 40.0
    y = ops.linear_fwd(x.cuda(), w.cuda(), None, 0).cpu()
    assert torch.equal(y, w.T.contiguous())
    y16 = ops.linear_fwd(x.cuda(), w.cuda(), None, 0, bf16=True).cpu()
    assert torch.equal(y16, w.T.contiguous())  # small integers are exact in bf16


@pytest.mark.parametrize("M,N,K", [(130, 32, 64), (513, 107, 214), (1024, 256, 432)])
def test_linear_bf16_within_1e2(ops, M, N, K):
    g = _gen(M + 1)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    y = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda(), 0, bf16=True).cpu()
    exp = torch.nn.functional.linear(x.double(), w.double(), b.double())
    assert_close(y.numpy(), exp.numpy(), rtol=1e-2, what="bf16 linear fwd", elem_rtol=False)
    dy = torch.randn(M, N, generator=g)
    dx = ops.linear_bwd_x(dy.cuda(), w.cuda(), bf16=True).cpu()
    assert_close(dx.numpy(), (dy.double() @ w.double()).numpy(), rtol=1e-2, what="bf16 bwd x", elem_rtol=False)
    dw, db = ops.linear_bwd_w(dy.cuda(), x.cuda(), bf16=True)
    assert_close(dw.cpu().numpy(), (dy.double().T @ x.double()).numpy(), rtol=1e-2, what="bf16 bwd w", elem_rtol=False)


@pytest.mark.parametrize("M,N,K", [(1000, 200, 136), (260, 132, 40), (4096, 1024, 432), (8192, 96, 72), (128, 128, 32),
                                   (2048, 216, 108), (1028, 108, 216)])
def test_linear_bf16_exact_on_rounded_operands(ops, M, N, K):
    """With operands that are already bf16 values the bf16-operand kernels (pipelined path:
    N > 64, aligned shapes) must match an fp32-accumulating reference to rounding: catches any
    tile, transposing-read or edge-masking mistake that a 1e-2 tolerance would hide."""
    g = _gen(M + N + K)
    r = lambda *sh: torch.randn(*sh, generator=g).bfloat16().float()
    x, w, b, dy = r(M, K), r(N, K) / 8, r(N), r(M, N)
    w = w.bfloat16().float()
    y = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda(), 1, bf16=True).cpu()
    exp = torch.nn.functional.linear(x.double(), w.double(), b.double()).relu()
    assert_close(y.numpy(), exp.numpy(), rtol=2e-6, what="bf16 fwd on rounded operands")
    dx = ops.linear_bwd_x(dy.cuda(), w.cuda(), bf16=True).cpu()
    assert_close(dx.numpy(), (dy.double() @ w.double()).numpy(), rtol=2e-6, what="bf16 bwd x on rounded operands")
    dw, db = ops.linear_bwd_w(dy.cuda(), x.cuda(), bf16=True)
    assert_close(dw.cpu().numpy(), (dy.double().T @ x.double()).numpy(), rtol=2e-5, what="bf16 bwd w on rounded operands")
    assert_close(db.cpu().numpy(), dy.double().sum(0).numpy(), rtol=2e-5, what="bf16 bwd b")


def test_linear_on_column_block(ops):
    g = _gen(2)
    wide = torch.randn(200, 96, generator=g).cuda()
    w = torch.randn(16, 32, generator=g).cuda()
    y = ops.linear_fwd(wide[:, 32:64], w, None, 0).cpu()
    assert_close(y.numpy(), (wide[:, 32:64].cpu().double() @ w.cpu().double().T).numpy(), what="column block")


# ------------------------------------------------------------------ loss
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_bce_matches_torch(ops, dtype):
    g = _gen(9)
    x = torch.randn(333, 5, generator=g) * 4
    y = (torch.rand(333, 5, generator=g) < 0.3).to(dtype)
    xr = x.clone().requires_grad_(True)
    ref = torch.nn.functional.binary_cross_entropy_with_logits(xr, y)
    ref.backward()
    loss, dl = ops.bce_with_logits(x.cuda(), y.cuda())
    assert loss.dtype == torch.float64
    assert_close(loss.item(), ref.item(), rtol=1e-6 if dtype == torch.float32 else 1e-9, what="bce loss")
    assert_close(dl.cpu().numpy(), xr.grad.numpy(), what="bce grad")


# ------------------------------------------------------------------ top-k, exchange helpers
@pytest.mark.parametrize("rows,n,k", [(37, 100, 10), (5, 64, 64), (3, 1000, 5), (1, 7, 1)])
def test_topk_matches_numpy_argsort(ops, rows, n, k):
    rng = np.random.RandomState(rows)
    scores = rng.randn(rows, n).astype(np.float32)
    codes = np.stack([rng.permutation(5000)[:n] for _ in range(rows)]).astype(np.int64)
    exp = np.take_along_axis(codes, np.argsort(scores, axis=1)[:, ::-1][:, :k], axis=1)
    got = ops.topk_codes(torch.from_numpy(scores).cuda(), torch.from_numpy(codes).cuda(), k).cpu().numpy()
    assert np.array_equal(got, exp)
    got_cols = ops.topk_codes(torch.from_numpy(scores).cuda(), None, k).cpu().numpy()
    assert np.array_equal(got_cols, np.argsort(scores, axis=1)[:, ::-1][:, :k])


@pytest.mark.parametrize("world", [1, 2, 8])
def test_bucketize_by_owner(ops, world):
    g = _gen(world)
    n, rows = 10000, 5000
    idx = torch.randint(0, rows, (n,), generator=g)
    counts, send_local, perm = ops.bucketize_by_owner(idx.cuda(), world, rows)
    owner = idx % world
    exp_perm = torch.sort(owner, stable=True)[1]
    assert torch.equal(perm.cpu().long(), exp_perm)
    assert torch.equal(counts.cpu(), torch.bincount(owner, minlength=world))
    assert torch.equal(send_local.cpu(), (idx // world)[exp_perm])
    rows_t = torch.randn(n, 16, generator=g)
    back = ops.unpermute_rows(rows_t.cuda(), perm).cpu()
    exp = torch.empty_like(rows_t)
    exp[exp_perm] = rows_t
    assert torch.equal(back, exp)
    assert torch.equal(ops.permute_rows(back.cuda(), perm).cpu(), rows_t)


@pytest.mark.parametrize("skew", ["uniform", "hot"])
@pytest.mark.parametrize("mode", ["dense", "sgd", "adagrad"])
def test_segreduce_companion_and_long_runs(ops, skew, mode):
    """Rows hit thousands of times (runs crossing many 32-position pieces) and the companion
    scalar table handled in the same launches."""
    from recsys_amd import _lib
    g = _gen(21)
    n, rows, dim, lr = 40000, 500, 16, 0.1
    if skew == "hot":
        idx = torch.where(torch.rand(n, generator=g) < 0.7, torch.randint(0, 3, (n,), generator=g),
                          torch.randint(0, rows, (n,), generator=g))
    else:
        idx = torch.randint(0, rows, (n,), generator=g)
    occ, occ2 = torch.randn(n, dim, generator=g), torch.randn(n, generator=g)
    si = ops.sort_dedup(idx.cuda(), rows)
    G = torch.zeros(rows, dim, dtype=torch.float64).index_add_(0, idx, occ.double())
    G2 = torch.zeros(rows, dtype=torch.float64).index_add_(0, idx, occ2.double())
    if mode == "dense":
        grad, grad2 = torch.zeros(rows, dim).cuda(), torch.zeros(rows, 1).cuda()
        sq = ops.segreduce(occ.cuda(), si, _lib.RX_SEG_DENSE, grad, occ2=occ2.cuda(), target2=grad2, want_sq=True)
        assert_close(grad.cpu().numpy(), G.numpy(), what="dense grad")
        assert_close(grad2.cpu().numpy()[:, 0], G2.numpy(), what="companion grad")
        assert_close(float(sq.double().sum()), float((G ** 2).sum() + (G2 ** 2).sum()), what="sum of squares")
        grad_b, grad2_b = torch.zeros(rows, dim).cuda(), torch.zeros(rows, 1).cuda()
        ops.segreduce(occ.cuda(), si, _lib.RX_SEG_DENSE, grad_b, occ2=occ2.cuda(), target2=grad2_b)
        assert torch.equal(grad, grad_b) and torch.equal(grad2, grad2_b)
        return
    w0, b0 = torch.randn(rows, dim, generator=g), torch.randn(rows, 1, generator=g)
    w, b = w0.clone().cuda(), b0.clone().cuda()
    if mode == "sgd":
        ops.segreduce(occ.cuda(), si, _lib.RX_SEG_SGD, w, lr=lr, occ2=occ2.cuda(), target2=b)
        assert_close(w.cpu().numpy(), (w0.double() - lr * G).numpy(), what="sgd rows")
        assert_close(b.cpu().numpy()[:, 0], (b0[:, 0].double() - lr * G2).numpy(), what="sgd companion")
    else:
        st, st2 = torch.zeros(rows, dim).cuda(), torch.zeros(rows, 1).cuda()
        ops.segreduce(occ.cuda(), si, _lib.RX_SEG_ADAGRAD, w, state=st, lr=lr, eps=1e-10, occ2=occ2.cuda(), target2=b, state2=st2)
        assert_close(st.cpu().numpy(), (G ** 2).numpy(), what="adagrad state")
        assert_close(w.cpu().numpy(), (w0.double() - lr * G / (G.abs() + 1e-10)).numpy(), rtol=1e-4, what="adagrad rows")
        assert_close(b.cpu().numpy()[:, 0], (b0[:, 0].double() - lr * G2 / (G2.abs() + 1e-10)).numpy(), rtol=1e-4, what="adagrad companion")


@pytest.mark.parametrize("mode", ["compact", "sgd", "adagrad"])
def test_multi_table_reduce_and_update_equal_one_launch_per_table_bitwise(ops, mode):
    """rx_segreduce_multi / rx_row_update_multi over tables of different heights, list lengths, lane
    groups (widths 64 and 32 share a layout class, 16-byte lanes with one step) and a companion on
    one of them: every output must equal the per-table launches bit for bit, partials included."""
    from recsys_amd import _lib
    g = _gen(77)
    shapes = [(5000, 700, 64, True), (1234, 90, 64, False), (40000, 3000, 32, False), (33, 5, 64, False)]
    lr, eps = 0.05, 1e-10
    tabs = []
    for (n, rows, dim, comp) in shapes:
        idx = torch.where(torch.rand(n, generator=g) < 0.3, torch.randint(0, 3, (n,), generator=g),
                          torch.randint(0, rows, (n,), generator=g)).cuda()
        tabs.append(dict(si=ops.sort_dedup(idx, rows), occ=torch.randn(n, dim, generator=g).cuda(),
                         occ2=torch.randn(n, generator=g).cuda() if comp else None,
                         w=torch.randn(rows, dim, generator=g).cuda(), b=torch.randn(rows, 1, generator=g).cuda() if comp else None,
                         n=n, rows=rows, dim=dim))
    scale = torch.tensor([0.37], device="cuda")
    M = {"compact": _lib.RX_SEG_COMPACT, "sgd": _lib.RX_SEG_SGD, "adagrad": _lib.RX_SEG_ADAGRAD}[mode]

    def fresh(t):
        return (t["w"].clone(), None if t["b"] is None else t["b"].clone(), torch.zeros_like(t["w"]),
                None if t["b"] is None else torch.zeros_like(t["b"]))

    if mode == "compact":
        single, items, multi = [], [], []
        for t in tabs:
            c, c2 = torch.zeros(t["n"], t["dim"]).cuda(), torch.zeros(t["n"]).cuda() if t["occ2"] is not None else None
            sq = ops.segreduce(t["occ"], t["si"], M, c, occ2=t["occ2"], target2=c2, want_sq=True)
            single.append((c, c2, sq))
            cm, cm2 = torch.zeros_like(c), None if c2 is None else torch.zeros_like(c2)
            sqm = torch.empty_like(sq)
            items.append(ops.TableReduce(t["occ"], t["si"], cm, t["dim"], occ2=t["occ2"], target2=cm2, sq=sqm))
            multi.append((cm, cm2, sqm))
        ops.segreduce_multi(items, M)
        for (c, c2, sq), (cm, cm2, sqm), t in zip(single, multi, tabs):
            nu = int(t["si"].n_uniq.item())
            assert torch.equal(c[:nu], cm[:nu]) and torch.equal(sq, sqm)
            assert c2 is None or torch.equal(c2[:nu], cm2[:nu])
        # the update from those compact gradients, one launch for all tables
        for rule in (_lib.RX_SEG_SGD, _lib.RX_SEG_ADAGRAD):
            ones, ups = [], []
            for (c, c2, _), t in zip(single, tabs):
                w, b, st, st2 = fresh(t)
                ops.row_update(c, t["si"], rule, w, state=st, lr=lr, eps=eps, scale=scale, compact2=c2, table2=b, state2=st2)
                ones.append((w, b, st, st2))
                wm, bm, stm, st2m = fresh(t)
                ups.append((ops.TableReduce(c, t["si"], wm, t["dim"], state=stm, occ2=c2, target2=bm, state2=st2m), (wm, bm, stm, st2m)))
            ops.row_update_multi([u[0] for u in ups], rule, lr=lr, eps=eps, scale=scale)
            for one, (_, many) in zip(ones, ups):
                for x, y in zip(one, many):
                    assert (x is None and y is None) or torch.equal(x, y)
        return
    ones, items, many = [], [], []
    for t in tabs:
        w, b, st, st2 = fresh(t)
        ops.segreduce(t["occ"], t["si"], M, w, state=st, lr=lr, eps=eps, scale=scale, occ2=t["occ2"], target2=b, state2=st2)
        ones.append((w, b, st, st2))
        wm, bm, stm, st2m = fresh(t)
        items.append(ops.TableReduce(t["occ"], t["si"], wm, t["dim"], state=stm, occ2=t["occ2"], target2=bm, state2=st2m))
        many.append((wm, bm, stm, st2m))
    ops.segreduce_multi(items, M, lr=lr, eps=eps, scale=scale)
    for one, m in zip(ones, many):
        for x, y in zip(one, m):
            assert (x is None and y is None) or torch.equal(x, y)
    # one table twice in a launch would have two writers per row: refused
    with pytest.raises(RuntimeError, match="once per call"):
        ops.segreduce_multi([items[0], items[0]], M, lr=lr, eps=eps)
    # widths of different lane layouts do not share a launch
    t = tabs[0]
    odd = ops.TableReduce(torch.randn(t["n"], 6).cuda(), t["si"], torch.zeros(t["rows"], 6).cuda(), 6,
                          state=torch.zeros(t["rows"], 6).cuda())
    with pytest.raises(RuntimeError, match="share the row shape"):
        ops.segreduce_multi([items[1], odd], M, lr=lr, eps=eps)


@pytest.mark.parametrize("mode", ["dense", "sgd", "adagrad", "compact"])
def test_segreduce_fm_matches_materialised_path(ops, mode):
    """rx_segreduce_fm (gradient synthesised from T and g) against rx_fm2_bwd + rx_segreduce."""
    from recsys_amd import _lib
    g0 = _gen(31)
    dims, dim, B = [4, 300, 9, 2000, 50], 16, 3000
    F = len(dims) - 1
    n_feat = sum(dims)
    feat = torch.randn(n_feat, dim, generator=g0).cuda()
    bias = torch.randn(n_feat, generator=g0).cuda()
    sparse = torch.stack([(torch.rand(B, generator=g0) ** 3 * d).long().clamp_(0, d - 1) for d in dims], 1).cuda()
    base = torch.tensor([0] + dims[:-1]).cumsum(0)
    ex_base, item_base = base[:F].cuda(), int(base[F])
    ex, item = sparse[:, :F], sparse[:, F:]
    logits, _, tsum = ops.fm2_fwd(feat, bias, ex, ex_base, item, item_base, want_tsum=True)
    g = torch.randn(B, 1, generator=g0).cuda() / B
    occ, bias_occ = ops.fm2_bwd(g, None, feat, bias, ex, ex_base, item, item_base)
    si = ops.sort_dedup(ops.fm2_flat_indices(feat, ex, ex_base, item, item_base), n_feat)
    lr = 0.2
    if mode == "dense":
        a, a2 = torch.zeros_like(feat), torch.zeros(n_feat, 1).cuda()
        b, b2 = torch.zeros_like(feat), torch.zeros(n_feat, 1).cuda()
        ops.segreduce(occ, si, _lib.RX_SEG_DENSE, a, occ2=bias_occ, target2=a2)
        ops.segreduce_fm(tsum, g.reshape(-1), F, si, _lib.RX_SEG_DENSE, b, table=feat, target2=b2)
    elif mode == "compact":
        a, a2 = torch.zeros(si.n, dim).cuda(), torch.zeros(si.n).cuda()
        b, b2 = torch.zeros(si.n, dim).cuda(), torch.zeros(si.n).cuda()
        sa = ops.segreduce(occ, si, _lib.RX_SEG_COMPACT, a, occ2=bias_occ, target2=a2, want_sq=True)
        sb = ops.segreduce_fm(tsum, g.reshape(-1), F, si, _lib.RX_SEG_COMPACT, b, table=feat, target2=b2, want_sq=True)
        assert_close(float(sb.double().sum()), float(sa.double().sum()), what="sum of squares")
        nu = int(si.n_uniq.item())
        a, a2, b, b2 = a[:nu], a2[:nu], b[:nu], b2[:nu]
    else:
        code = _lib.RX_SEG_SGD if mode == "sgd" else _lib.RX_SEG_ADAGRAD
        a, a2, b, b2 = feat.clone(), bias.clone().reshape(-1, 1), feat.clone(), bias.clone().reshape(-1, 1)
        st = [torch.zeros_like(x) for x in (a, a2, b, b2)] if mode == "adagrad" else [None] * 4
        ops.segreduce(occ, si, code, a, state=st[0], lr=lr, eps=1e-10, occ2=bias_occ, target2=a2, state2=st[1])
        ops.segreduce_fm(tsum, g.reshape(-1), F, si, code, b, state=st[2], lr=lr, eps=1e-10, target2=b2, state2=st[3])
    rt = 1e-4 if mode == "adagrad" else 1e-5
    assert_close(b.cpu().numpy(), a.cpu().numpy(), rtol=rt, what=f"{mode} rows", atol=1e-9)
    assert_close(b2.cpu().numpy(), a2.cpu().numpy(), rtol=rt, what=f"{mode} bias", atol=1e-9)


def test_bce_gradient_at_exactly_zero_logits(ops):
    """sigmoid(0) = 0.5: the logit gradient there is (0.5 - y) / M, as torch's op gives."""
    g = _gen(3)
    x = torch.randn(64, 3, generator=g)
    x[::7] = 0.0
    y = (torch.rand(64, 3, generator=g) < 0.4).double()
    loss, dl = ops.bce_with_logits(x.cuda(), y.cuda())
    xr = x.clone().requires_grad_(True)
    ref = torch.nn.functional.binary_cross_entropy_with_logits(xr, y)
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-7 * abs(ref.item())
    assert torch.allclose(dl.cpu(), xr.grad, rtol=1e-6, atol=0)


# ------------------------------------------------------------------ dense layers with bf16 activations in HBM
def _r16(t):
    return t.bfloat16().float()


def test_to_bf16_and_transposed_copy(ops):
    g = _gen(3)
    x = torch.randn(301, 432, generator=g).cuda()
    assert torch.equal(ops.to_bf16(x), x.bfloat16())
    w = torch.randn(107, 216, generator=g).cuda()
    assert torch.equal(ops.to_bf16(w, transpose=True), w.t().contiguous().bfloat16())
    view = torch.randn(64, 440, generator=g).cuda()[:, :432]          # a column block of a wider matrix
    assert torch.equal(ops.to_bf16(view), view.bfloat16())


def _dev16(t):
    """A bf16 matrix on the device the way the kernels want it: rows padded with zeros to 8 elements."""
    from recsys_amd import ops as o
    out = o._padded16(t.shape[0], t.shape[1], "cuda")
    out.copy_(t)
    return out


@pytest.mark.parametrize("M,N,K", [(1000, 432, 432), (513, 104, 216), (4096, 1024, 432), (130, 256, 512), (129, 8, 8),
                                   (300, 1, 256), (4097, 1, 64), (777, 108, 216), (777, 216, 108), (300, 1, 108), (200, 12, 20)])
def test_linear16_matches_products_of_the_rounded_operands(ops, M, N, K):
    """Forward (bf16 and fp32 output, bias, ReLU), input gradient (fp32, accumulated, and bf16 with the
    ReLU gate) and weight gradient + column sums against fp64 products of the bf16-rounded operands;
    widths that are multiples of 4 but not of 8 ride on the zero padding of the bf16 rows."""
    g = _gen(M + N + K)
    x16 = torch.randn(M, K, generator=g).bfloat16()
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ref = x16.double() @ _r16(w).double().t() + b.double()
    y32 = ops.linear16_fwd(_dev16(x16), w.cuda(), b.cuda(), relu=False, out_bf16=False)
    assert y32.dtype == torch.float32
    assert_close(y32.cpu().numpy(), ref.numpy(), rtol=1e-5, what="linear16 fwd fp32")
    if N > 1:
        y16 = ops.linear16_fwd(_dev16(x16), w.cuda(), b.cuda(), relu=True, out_bf16=True)
        assert y16.dtype == torch.bfloat16
        # one bf16 rounding of the fp32 result: within half a bf16 ulp (2^-9) of the exact value
        assert_close(y16.float().cpu().numpy(), ref.relu().numpy(), rtol=2.0 ** -8, what="linear16 fwd bf16", elem_rtol=False)
        assert torch.equal(y16.cpu(), ops.linear16_fwd(_dev16(x16), w.cuda(), b.cuda(), relu=True, out_bf16=False).bfloat16().cpu())
        assert not y16.as_strided((M, (N + 7) // 8 * 8), (y16.stride(0), 1))[:, N:].any(), "row padding must stay zero"
    dy16 = torch.randn(M, N, generator=g).bfloat16()
    refx = dy16.double() @ _r16(w).double()
    if N > 1:
        dx = ops.linear16_bwd_x(_dev16(dy16), w.cuda(), out_bf16=False)
        assert_close(dx.cpu().numpy(), refx.numpy(), rtol=1e-5, what="linear16 bwd x")
        base = torch.randn(M, K, generator=g)
        acc = base.clone().cuda()
        ops.linear16_bwd_x(_dev16(dy16), w.cuda(), out=acc, accumulate=True)
        assert_close(acc.cpu().numpy(), (base.double() + refx).numpy(), rtol=1e-5, what="linear16 bwd x accumulated")
    gate = torch.randn(M, K, generator=g).relu().bfloat16()
    gate[0, 0] = -0.0
    dz = ops.linear16_bwd_x(_dev16(dy16), w.cuda(), gate16=_dev16(gate), out_bf16=True)
    expz = torch.where(gate.float() > 0, refx, torch.zeros_like(refx))
    assert dz.dtype == torch.bfloat16
    assert_close(dz.float().cpu().numpy(), expz.numpy(), rtol=2.0 ** -8, what="linear16 bwd x gated", elem_rtol=False)
    assert not dz.float().cpu()[gate.float() <= 0].any()
    assert not dz.as_strided((M, (K + 7) // 8 * 8), (dz.stride(0), 1))[:, K:].any(), "row padding must stay zero"
    dw, db = ops.linear16_bwd_w(_dev16(dy16), _dev16(x16), True)
    assert_close(dw.cpu().numpy(), (dy16.double().t() @ x16.double()).numpy(), rtol=1e-5, what="linear16 bwd w")
    assert_close(db.cpu().numpy(), dy16.double().sum(0).numpy(), rtol=1e-5, what="linear16 bias grad")
    dw2, _ = ops.linear16_bwd_w(_dev16(dy16), _dev16(x16), True)
    assert torch.equal(dw, dw2)


def test_linear16_tile_layout_asymmetric(ops):
    """A = I against an asymmetric integer-valued B: exact in bf16, so a transposed or permuted
    accumulator write in the LDS-staged epilogue shows as a wrong entry, in every output form."""
    n = 200
    x16 = _dev16(torch.eye(n).bfloat16())
    w = ((torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 97) - 40.0).cuda()
    assert torch.equal(ops.linear16_fwd(x16, w, None, relu=False, out_bf16=False), w.t())
    assert torch.equal(ops.linear16_fwd(x16, w, None, relu=False, out_bf16=True).float(), w.t())
    assert torch.equal(ops.linear16_bwd_x(x16, w, out_bf16=False), w)                      # dX = I W
    dw, _ = ops.linear16_bwd_w(_dev16(w.bfloat16().cpu()), x16, False)                     # dW = W^T I
    assert torch.equal(dw, w.t())


def test_cross16_combine_matches_torch(ops):
    g = _gen(11)
    rows, cols = 777, 432
    x0, u, xl, d = (torch.randn(rows, cols, generator=g).cuda() for _ in range(4))
    xn, xn16 = ops.cross16_fwd(x0, u, xl)
    exp = x0 * u + xl
    assert torch.equal(xn, exp) and torch.equal(xn16, exp.bfloat16())
    for accumulate, identity, want in ((False, False, True), (True, False, False), (True, True, False)):
        acc0 = torch.randn(rows, cols, generator=g).cuda()
        acc = acc0.clone()
        du16, dxl = ops.cross16_bwd(d, x0, u, acc, accumulate=accumulate, identity=identity, want_dxl=want)
        assert torch.equal(du16, (d * x0).bfloat16())
        g0 = d * u
        if identity:
            g0 = g0 + d
        if accumulate:
            g0 = g0 + acc0
        assert torch.equal(acc, g0)
        assert (dxl is None) == (not want) and (dxl is None or torch.equal(dxl, d))


@pytest.mark.parametrize("M,d", [(1000, 432), (129, 176), (4096, 64)])
def test_cross16_layer_in_one_launch_equals_product_then_combine(ops, M, d):
    g = _gen(M + d)
    x0, xl = torch.randn(M, d, generator=g).cuda(), torch.randn(M, d, generator=g).cuda()
    w, b = (torch.randn(d, d, generator=g) / d ** 0.5).cuda(), torch.randn(d, generator=g).cuda()
    x16 = ops.to_bf16(xl)
    u, xn, xn16 = ops.cross16_layer_fwd(x16, w, b, x0, xl)
    u2 = ops.linear16_fwd(x16, w, b, relu=False, out_bf16=False)
    xn2, xn216 = ops.cross16_fwd(x0, u2, xl)
    assert torch.equal(u, u2) and torch.equal(xn, xn2) and torch.equal(xn16, xn216)


# ------------------------------------------------------------------ Linear + ReLU + Dropout in one launch
def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("M,N,K", [(1000, 64, 128), (300, 24, 40), (3000, 432, 432)])
def test_linear_relu_dropout_node_matches_torch_composite(ops, M, N, K):
    """autograd.LinearAct with a dropout rate: torch's dropout forward on the fused Linear + ReLU, and ONE
    backward pass over the pair's output.  Reseeding reproduces the mask: output and the gradients of
    x, W, b against the same expression in torch ops."""
    from recsys_amd import autograd as rxa
    g = _gen(M + N)
    p = 0.3
    x = torch.randn(M, K, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().requires_grad_(True)
    b = torch.randn(N, generator=g).cuda().requires_grad_(True)
    dy = torch.randn(M, N, generator=g).cuda()
    torch.manual_seed(77)
    y = rxa.linear(x, w, b, relu=True, drop_p=p)
    y.backward(dy)
    got = [y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone()]
    for t in (x, w, b):
        t.grad = None
    torch.manual_seed(77)
    ref = torch.nn.functional.dropout(torch.relu(torch.nn.functional.linear(x, w, b)), p, True)
    ref.backward(dy)
    assert_close(_np(got[0]), _np(ref), rtol=1e-5, what="dropout node output")
    assert int(((got[0] == 0) != (ref.detach() == 0)).sum()) <= (M * N) // 1000     # a pre-activation within rounding of zero may flip
    assert 0.55 < float((got[0] == 0).float().mean()) < 0.75                          # half closed by the ReLU, 30 % of the rest dropped
    assert_close(_np(got[1]), _np(x.grad), rtol=1e-4, what="dropout node grad x")
    assert_close(_np(got[2]), _np(w.grad), rtol=1e-4, what="dropout node grad w")
    assert_close(_np(got[3]), _np(b.grad), rtol=1e-4, what="dropout node grad b")


def test_mlp_tower_under_dropout_matches_composite(ops):
    """The MLP model in training mode (Linear-ReLU-Dropout twice, p = 0.5): the masks come from torch's device
    generator in layer order, so reseeding reproduces them; logits and the tower's weight gradients equal
    the composite of torch ops.  In eval mode the Dropouts are idle."""
    from recsys_amd.models.neuralcf import MLP
    torch.manual_seed(0)
    m = MLP(50, 40, 16, dropout=0.5).cuda().train()
    users = torch.randint(0, 50, (64,)).cuda()
    items = torch.randint(0, 40, (64, 5)).cuda()
    batch = {"user_code": users, "item_code": items}
    torch.manual_seed(123)
    out = m(batch)
    out.sum().backward()
    torch.manual_seed(123)
    lin = [l for l in m.mlp if isinstance(l, torch.nn.Linear)]
    W = [l.weight.detach().clone().requires_grad_(True) for l in lin]
    Bv = [l.bias.detach() for l in lin]
    x = torch.cat([m.user_embedding.weight.detach()[users].repeat_interleave(5, 0), m.item_embedding.weight.detach()[items.reshape(-1)]], 1)
    h = torch.nn.functional.dropout(torch.relu(x @ W[0].t() + Bv[0]), 0.5, True)
    h = torch.nn.functional.dropout(torch.relu(h @ W[1].t() + Bv[1]), 0.5, True)
    ref = (h @ W[2].t() + Bv[2]).view(64, 5)
    ref.sum().backward()
    assert_close(_np(out.detach()), _np(ref.detach().float()), rtol=1e-5, what="MLP logits under dropout")
    for l, wr in zip(lin, W):
        assert_close(_np(l.weight.grad), _np(wr.grad.float()), rtol=1e-4, what="MLP weight grad under dropout")
    m.eval()
    a, b2 = m(batch), m(batch)
    assert torch.equal(a, b2)


def test_zero_unused_slots_clears_the_padding_and_nothing_else(ops):
    """rx_zero_unused_slots on a [G * capacity, dim] send buffer: slots at offset >= counts[owner] become
    zero (rows and companion entries), used slots keep their bits; a full or overfull group is left alone."""
    G, cap = 5, 37
    counts = torch.tensor([0, 11, 37, 50, 36]).cuda()
    for dim in (4, 16, 128):
        rows = torch.randn(G * cap, dim, generator=_gen(dim)).cuda()
        comp = torch.randn(G * cap, generator=_gen(dim + 1)).cuda()
        want, want2 = rows.clone(), comp.clone()
        for o in range(G):
            want[o * cap + min(int(counts[o]), cap):(o + 1) * cap] = 0
            want2[o * cap + min(int(counts[o]), cap):(o + 1) * cap] = 0
        ops.zero_unused_slots(rows, comp, counts, cap)
        assert torch.equal(rows, want) and torch.equal(comp, want2)
        rows2 = want.clone().fill_(1.0)
        ops.zero_unused_slots(rows2, None, counts, cap)
        assert int((rows2 == 0).all(1).sum()) == sum(cap - min(int(c), cap) for c in counts)
    with pytest.raises(ValueError):
        ops.zero_unused_slots(torch.zeros(7, 4).cuda(), None, counts, cap)


def test_index_glue_kernels_equal_their_torch_formulas(ops):
    """rx_fm_flat_rows against add + slice + cat, rx_compose_slots against widen + index; in place too."""
    g = _gen(21)
    dims = [7, 1000, 13, 5000, 3, 211]
    offs = torch.tensor([0] + list(torch.tensor(dims).cumsum(0)[:-1])).cuda()
    for B in (1, 300, 4097):
        sparse = torch.stack([torch.randint(0, d, (B,), generator=g) for d in dims], 1).cuda()
        glob = sparse + offs[None, :]
        want = torch.cat([glob[:, :-1].reshape(-1), glob[:, -1]])
        got = ops.fm_flat_rows(sparse, offs)
        assert torch.equal(got, want)
        again = ops.fm_flat_rows(sparse.flip(0).contiguous(), offs, out=got)          # in place
        flipped = glob.flip(0)
        assert again is got and torch.equal(got, torch.cat([flipped[:, :-1].reshape(-1), flipped[:, -1]]))
    src_of = torch.randint(-1, 900, (500,), generator=g, dtype=torch.int32).cuda()
    inverse = torch.randint(0, 500, (3000,), generator=g, dtype=torch.int32).cuda()
    slot_of, occ = ops.compose_slots(src_of, inverse)
    assert slot_of.dtype == occ.dtype == torch.int64
    assert torch.equal(slot_of, src_of.long()) and torch.equal(occ, src_of.long()[inverse.long()])
    out = (torch.zeros_like(slot_of), torch.zeros_like(occ))
    assert ops.compose_slots(src_of, inverse, out=out) is out and torch.equal(out[1], occ)
    with pytest.raises(ValueError):
        ops.compose_slots(src_of, inverse, out=(out[0][:10], out[1]))
    with pytest.raises(ValueError):
        ops.fm_flat_rows(sparse, offs[:3])


def test_multi_dense_adam_equals_one_launch_per_tensor_bitwise(ops):
    """rx_multi_dense_adam: torch.optim.Adam's step over many tensors in one launch; bit for bit what
    rx_dense_adam / rx_dense_adam_dstep leave tensor by tensor, with the step count by value or on the device."""
    g0 = _gen(77)
    shapes = [(6040, 64), (3706, 64), (64, 128), (64,), (1,), (33, 7), (32, 64)]
    scale = torch.tensor([0.37], device="cuda")
    for dstep in (False, True):
        ws = [torch.randn(s, generator=g0).cuda() for s in shapes]
        gs = [torch.randn(s, generator=g0).cuda() for s in shapes]
        ms = [torch.randn(s, generator=g0).cuda() * 0.1 for s in shapes]
        vs = [torch.rand(s, generator=g0).cuda() * 0.01 for s in shapes]
        ref = [[t.clone() for t in ts] for ts in (ws, ms, vs)]
        step = torch.tensor([3], dtype=torch.int32, device="cuda") if dstep else 3
        for w, g, m, v in zip(ref[0], gs, ref[1], ref[2]):
            ops.dense_adam(w, g, m, v, 1e-3, 0.9, 0.999, 1e-8, step, scale)
        ops.multi_dense_adam(ws, gs, ms, vs, 1e-3, 0.9, 0.999, 1e-8, step, scale)
        for a, b in zip(ws + ms + vs, ref[0] + ref[1] + ref[2]):
            assert torch.equal(a, b)
        # and against torch's own formula for one of them (fp32 tolerance)
        w0 = torch.randn(shapes[2], generator=g0).cuda()
        p = torch.nn.Parameter(w0.clone())
        opt = torch.optim.Adam([p], lr=1e-3)
        m0, v0 = torch.zeros_like(w0), torch.zeros_like(w0)
        for t in (1, 2):
            g = torch.randn(shapes[2], generator=g0).cuda()
            p.grad = g.clone()
            opt.step()
            ops.multi_dense_adam([w0], [g], [m0], [v0], 1e-3, 0.9, 0.999, 1e-8,
                                 torch.tensor([t], dtype=torch.int32, device="cuda") if dstep else t)
        assert_close(w0.cpu(), p.detach().cpu(), 1e-5, "multi adam vs torch.optim.Adam")


def test_dense_gradients_of_several_tables_in_one_reduce_equal_one_reduce_per_table(ops):
    """autograd.tables_backward: four dense table gradients out of one zeroed buffer and one pair of reduce
    launches; the same bits as table_backward table by table (and as index_add in fp64, to tolerance)."""
    from recsys_amd import autograd as rxa
    g0 = _gen(5)
    U, I, D, B, N = 600, 371, 64, 900, 6
    tabs = [torch.nn.Parameter(torch.randn(n, D, generator=g0).cuda()) for n in (U, I, U, I)]
    users = torch.randint(0, U, (B,), generator=g0).cuda()
    items = torch.randint(0, I, (B, N), generator=g0).cuda()
    su, si = rxa.sorted_index(users, U), rxa.sorted_index(items, I)
    occs = [torch.randn(B, D, generator=g0).cuda(), torch.randn(B * N, D, generator=g0).cuda(),
            torch.randn(B, D, generator=g0).cuda(), torch.randn(B * N, D, generator=g0).cuda()]
    sis = (su, si, su, si)
    together = rxa.tables_backward(tabs, sis, occs)
    for p, s, occ, g in zip(tabs, sis, occs, together):
        alone = rxa.table_backward(p, s, occ)
        assert g.shape == p.shape and torch.equal(g, alone)
        idx = users if occ.shape[0] == B else items.reshape(-1)
        exp = torch.zeros(p.shape, dtype=torch.float64, device="cuda").index_add_(0, idx, occ.double())
        assert_close(g.cpu(), exp.float().cpu(), 1e-5, "dense table gradient")


@pytest.mark.parametrize("K", [520, 836, 2048, 4096])
def test_linear16_one_column_layer_beyond_512_inputs(ops, K):
    """The tower's last layer (one output column) on bf16 activations with rows of more than 512 inputs (the
    836-wide last layer of the 1 B-row DeepFM's tower): forward, input gradient and weight / bias gradient against
    fp64 products of the same bf16-rounded operands."""
    g0 = _gen(K)
    M = 3001
    x = torch.randn(M, K, generator=g0).cuda()
    w = (torch.randn(1, K, generator=g0) * 0.1).cuda()
    b = torch.randn(1, generator=g0).cuda()
    dy = torch.randn(M, 1, generator=g0).cuda()
    x16, dy16 = ops.to_bf16(x), ops.to_bf16(dy)
    xr, wr, dyr = x16.float().double(), w.to(torch.bfloat16).double(), dy16.float().double()
    y = ops.linear16_fwd(x16, w, b, False, False)
    assert_close(y.cpu(), (xr @ wr.t() + b.double()).float().cpu(), 1e-5, "linear16 one-column forward")
    dx = ops.linear16_bwd_x(dy16, w, out_bf16=False)
    assert_close(dx.cpu(), (dyr @ wr).float().cpu(), 1e-5, "linear16 one-column input gradient")
    dw, db = ops.linear16_bwd_w(dy16, x16, True)
    assert_close(dw.cpu(), (dyr.t() @ xr).float().cpu(), 1e-5, "linear16 one-column weight gradient")
    assert_close(db.cpu(), dyr.sum(0).float().cpu(), 1e-5, "linear16 bias grad")
