"""Field-grouped backward (rx_fg_group / rx_fg_reduce / rx_fg_row_update) against torch restatements:
the grouping order bit for bit, the distinct-row set against torch.unique, gradients and updated rows
against an fp64 scatter-add."""
import numpy as np
import pytest
import torch

from conftest import assert_close as _assert_close

pytestmark = pytest.mark.gpu


def assert_close(a, e, **kw):
    _assert_close(a.detach().cpu().numpy(), e.detach().cpu().numpy(), **kw)


def _bases(dims):
    out = [0]
    for d in dims:
        out.append(out[-1] + int(d))
    return out


def _expected_order(sparse, dims):
    """Per field: positions ordered by (bin, row >> 8, sample), bin = row for the field's first 256
    rows and 256 + (row & 255) beyond — numpy lexsort, stable."""
    B, nf = sparse.shape
    base = _bases(dims)
    srow, sb = [], []
    s = sparse.cpu().numpy()
    for f in range(nf):
        r = s[:, f]
        order = np.lexsort((np.arange(B), r >> 8, np.where(r < 256, r, 256 + (r & 255))))
        srow.append(r[order] + base[f])
        sb.append(order)
    return np.concatenate(srow), np.concatenate(sb)


def _group(sparse, dims):
    from recsys_amd import ops
    F = sparse.shape[1] - 1
    return ops.fg_group(sparse[:, :F], sparse[:, F:], _bases(dims))


CASES = [
    ([7, 1000, 13, 5000, 3, 211], 300, "zipf"),
    ([300, 70000, 257, 256, 2, 1 << 20], 4096 + 17, "zipf"),
    ([5, 40000], 1, "uniform"),
    ([100000, 9, 3000000], 9000, "uniform"),
    ([1 << 24, 50], 20000, "uniform"),
    ([600, 90000], 30000, "one_row"),          # a single row (id >= 256) holds the whole field: beyond the small bucket kernel
    # fields whose (row >> 8, sample) words need more than 32 bits: the 64-bit instantiation
    ([1 << 26, 50], 20000, "uniform"),
    ([300, 50_000_000, 257, 1 << 27], 4096 + 17, "zipf"),
    ([600, 1 << 25], 40000, "one_row"),
]


@pytest.mark.parametrize("dims,B,dist", CASES)
def test_grouping_order_is_exact(dims, B, dist):
    from recsys_amd import ops, synthetic
    g = torch.Generator(device="cuda").manual_seed(B)
    if dist == "zipf":
        sparse = synthetic.criteo_batch(dims, B, seed=B, device="cuda")["sparse"]
    elif dist == "uniform":
        sparse = torch.stack([torch.randint(0, d, (B,), generator=g, device="cuda") for d in dims], 1).contiguous()
    else:
        sparse = torch.stack([torch.full((B,), min(d - 1, 300 + 7 * k), device="cuda", dtype=torch.int64) for k, d in enumerate(dims)], 1).contiguous()
    fg = _group(sparse, dims)
    ops.raise_if_index_error()
    exp_row, exp_b = _expected_order(sparse, dims)
    got_row = (fg.srow.to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
    got_b = (fg.sb.to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
    assert np.array_equal(got_row, exp_row)
    assert np.array_equal(got_b, exp_b)
    flat = (sparse + torch.tensor(_bases(dims)[:-1], device="cuda")[None, :]).reshape(-1)
    assert torch.equal(torch.sort(fg.distinct_rows()).values, torch.unique(flat))


def test_grouping_at_the_largest_supported_batch_and_a_quarter_batch_row():
    """B = 65536 with Criteo-like fields; one field has a row that a third of the samples name and
    that sits above id 255, so its bucket exceeds the LDS budget and is sorted through global memory."""
    from recsys_amd import ops, synthetic
    dims = [26361, 262552, 3483, 8129362, 27, 9995165, 70000]
    B = 65536
    sparse = synthetic.criteo_batch(dims, B, seed=3, device="cuda")["sparse"]
    hot = torch.rand(B, device="cuda") < 0.34
    sparse[hot, 6] = 40001
    fg = _group(sparse, dims)
    ops.raise_if_index_error()
    exp_row, exp_b = _expected_order(sparse, dims)
    assert np.array_equal((fg.srow.to(torch.int64) & 0xFFFFFFFF).cpu().numpy(), exp_row)
    assert np.array_equal((fg.sb.to(torch.int64) & 0xFFFFFFFF).cpu().numpy(), exp_b)
    again = _group(sparse, dims)
    assert torch.equal(again.srow, fg.srow) and torch.equal(again.sb, fg.sb)


def test_grouping_of_125m_row_tables_at_the_largest_batch():
    """The field mix of the 125 M-row x 128 workloads at B = 65536 (34-bit words: the 64-bit
    instantiation), with one row above id 255 named by a third of the samples (sorted through global
    memory) and one named by 3000 of them (the LDS-resident big-bin path with 64-bit words)."""
    from recsys_amd import ops, synthetic
    dims = synthetic.criteo_field_dims(total_rows=125_000_000)
    B = 65536
    sparse = synthetic.criteo_batch(dims, B, seed=5, device="cuda")["sparse"]
    big = int(np.argmax(dims))
    assert dims[big] > (1 << 24)
    r = torch.rand(B, device="cuda")
    sparse[r < 0.34, big] = 40001
    sparse[(r >= 0.34) & (r < 0.34 + 3000 / B), big] = 12_345_678
    fg = _group(sparse, dims)
    ops.raise_if_index_error()
    exp_row, exp_b = _expected_order(sparse, dims)
    assert np.array_equal((fg.srow.to(torch.int64) & 0xFFFFFFFF).cpu().numpy(), exp_row)
    assert np.array_equal((fg.sb.to(torch.int64) & 0xFFFFFFFF).cpu().numpy(), exp_b)


def test_grouping_flags_an_index_outside_its_field():
    from recsys_amd import ops
    dims = [10, 300]
    sparse = torch.tensor([[1, 5], [10, 7], [3, 299]], device="cuda")    # 10 is one past field 0
    ops.set_index_check("deferred")
    try:
        _group(sparse, dims)
        with pytest.raises(IndexError):
            ops.raise_if_index_error()
    finally:
        ops.set_index_check("sync")


def test_a_flagged_step_writes_only_rows_of_the_batch_or_the_clamped_row():
    """An index outside its field is flagged (IndexError when the flag is read).  The forward leaves the
    occurrence out; the index preparation clamps it into its field, so the backward adds that occurrence's
    gradient to the field's nearest row.  What the contract promises about the tables after such a step:
    nothing is written outside the rows the batch names plus that clamped row."""
    from recsys_amd import ops, optim
    from recsys_amd.models import fields
    dims = [10, 300, 40]
    base = _bases(dims)
    sparse = torch.tensor([[1, 5, 0], [12, 7, 39], [3, 299, 8], [-2, 0, 8]], device="cuda")   # 12 and -2 are outside field 0
    torch.manual_seed(0)
    with torch.device("cuda"):
        m = fields.FieldFM(dims, 16).train()
    w0, b0 = m.feature_embedding.weight.detach().clone(), m.bias_embedding.weight.detach().clone()
    opt = optim.FusedOptimizer(m, kind="sgd", lr=0.1, clip_norm=1.0)
    ops.set_index_check("deferred")
    try:
        opt.zero_grad()
        m.loss({"sparse": sparse, "label": torch.tensor([1., 0., 1., 0.], device="cuda", dtype=torch.float64)}).backward()
        opt.step()
        with pytest.raises(IndexError):
            ops.raise_if_index_error()
    finally:
        ops.set_index_check("sync")
    allowed = torch.zeros(sum(dims), dtype=torch.bool, device="cuda")
    valid = (sparse >= 0) & (sparse < torch.tensor(dims, device="cuda")[None, :])
    rows = sparse + torch.tensor(base[:-1], device="cuda")[None, :]
    allowed[rows[valid]] = True
    allowed[base[0]] = True             # -2 clamps to the field's first row
    allowed[base[1] - 1] = True         # 12 clamps to its last
    w1, b1 = m.feature_embedding.weight.detach(), m.bias_embedding.weight.detach()
    assert torch.equal(w1[~allowed], w0[~allowed]) and torch.equal(b1[~allowed], b0[~allowed])
    assert torch.isfinite(w1).all() and torch.isfinite(b1).all()


def _reference_grads(table, bias, sparse, dims, T, g, deep, dim):
    """fp64: G_r = sum_b g_b (T_b - e_r) + tower slices; bias gradient sum_b g_b."""
    B, nf = sparse.shape
    base = torch.tensor(_bases(dims)[:-1], device=sparse.device)
    rows = sparse + base[None, :]
    G = torch.zeros(table.shape, dtype=torch.float64, device=table.device)
    G2 = torch.zeros(table.shape[0], dtype=torch.float64, device=table.device)
    for f in range(nf):
        r = rows[:, f]
        occ = g.double()[:, None] * (T.double() - table.double()[r])
        if deep is not None:
            sl = 0 if f == nf - 1 else 1 + f
            occ = occ + deep.double()[:, sl * dim:(sl + 1) * dim]
        G.index_add_(0, r, occ)
        G2.index_add_(0, r, g.double())
    return G, G2


@pytest.mark.parametrize("dim", [16, 32, 64, 128])
@pytest.mark.parametrize("with_deep", [False, True])
def test_reduce_modes_match_an_fp64_scatter_add(dim, with_deep):
    from recsys_amd import _lib, ops, synthetic
    dims = [7, 1000, 13, 5000, 3, 211, 40000]
    B = 1500
    torch.manual_seed(dim)
    nf = len(dims)
    n_rows = sum(dims)
    table = torch.randn(n_rows, dim, device="cuda")
    bias = torch.randn(n_rows, device="cuda")
    sparse = synthetic.criteo_batch(dims, B, seed=dim, device="cuda")["sparse"]
    T = torch.randn(B, dim, device="cuda")
    g = torch.randn(B, device="cuda") / B
    deep = (torch.randn(B, nf * dim + 12, device="cuda") / B) if with_deep else None
    fg = _group(sparse, dims)
    Gref, G2ref = _reference_grads(table, bias, sparse, dims, T, g, deep, dim)

    # dense gradients
    grad, grad2 = torch.zeros_like(table), torch.zeros_like(bias)
    ops.fg_reduce(T, g, fg, _lib.RX_SEG_DENSE, grad, table=table, target2=grad2, deep=deep)
    assert_close(grad, Gref.float(), rtol=1e-5, what="dense grad")
    assert_close(grad2, G2ref.float(), rtol=1e-5, what="dense bias grad")
    again, again2 = torch.zeros_like(table), torch.zeros_like(bias)
    ops.fg_reduce(T, g, fg, _lib.RX_SEG_DENSE, again, table=table, target2=again2, deep=deep)
    assert torch.equal(again, grad) and torch.equal(again2, grad2)          # run-to-run bitwise

    # compact gradients at the runs' last positions + sum of squares
    comp = torch.full((fg.n, dim), float("nan"), device="cuda")
    comp2 = torch.full((fg.n,), float("nan"), device="cuda")
    sq = ops.fg_reduce(T, g, fg, _lib.RX_SEG_COMPACT, comp, table=table, target2=comp2, want_sq=True, deep=deep)
    r = fg.srow.to(torch.int64) & 0xFFFFFFFF
    end = torch.ones_like(r, dtype=torch.bool)
    end[:-1] = r[:-1] != r[1:]
    assert torch.equal(comp[end], grad[r[end]]) and torch.equal(comp2[end], grad2[r[end]])
    total = float(sq.double().sum().item())
    ref_total = float((Gref ** 2).sum().item() + (G2ref ** 2).sum().item())
    assert abs(total - ref_total) <= 1e-5 * ref_total

    # fused SGD in place == dense SGD on the touched rows, untouched rows bit-identical
    lr = 0.3
    w, w2 = table.clone(), bias.clone()
    ops.fg_reduce(T, g, fg, _lib.RX_SEG_SGD, w, target2=w2, lr=lr, deep=deep)
    assert_close(w, (table.double() - lr * Gref).float(), rtol=1e-5, what="sgd rows")
    assert_close(w2, (bias.double() - lr * G2ref).float(), rtol=1e-5, what="sgd bias")
    touched = torch.zeros(n_rows, dtype=torch.bool, device="cuda")
    touched[r] = True
    assert torch.equal(w[~touched], table[~touched]) and torch.equal(w2[~touched], bias[~touched])

    # two-phase: compact + row update with a scale == one-phase with the same scale
    scale = torch.tensor([0.37], device="cuda")
    for mode, name in ((_lib.RX_SEG_SGD, "sgd"), (_lib.RX_SEG_ADAGRAD, "adagrad"), (_lib.RX_SEG_ROWWISE_ADAGRAD, "rowwise")):
        shape = (n_rows, dim) if mode == _lib.RX_SEG_ADAGRAD else (n_rows,)
        st_a = torch.rand(shape, device="cuda") if mode != _lib.RX_SEG_SGD else None
        st2_a = torch.rand(n_rows, device="cuda") if mode != _lib.RX_SEG_SGD else None
        st_b = st_a.clone() if st_a is not None else None
        st2_b = st2_a.clone() if st2_a is not None else None
        wa, wa2 = table.clone(), bias.clone()
        ops.fg_reduce(T, g, fg, mode, wa, state=st_a, target2=wa2, state2=st2_a, lr=lr, eps=1e-10, scale=scale, deep=deep)
        wb, wb2 = table.clone(), bias.clone()
        ops.fg_row_update(comp, fg, mode, wb, state=st_b, lr=lr, eps=1e-10, scale=scale, compact2=comp2, table2=wb2, state2=st2_b)
        assert torch.equal(wa, wb) and torch.equal(wa2, wb2), name
        if st_a is not None:
            assert torch.equal(st_a, st_b) and torch.equal(st2_a, st2_b), name
        assert torch.equal(wa[~touched], table[~touched]), name
        if mode == _lib.RX_SEG_SGD:
            assert_close(wa, (table.double() - lr * 0.37 * Gref).float(), rtol=1e-5, what="scaled sgd")

    # speculative clipped SGD: the reduce takes the step with factor 1 and leaves undo records; the row
    # update ends at once for a factor of exactly 1 and otherwise redoes the rows -- bit for bit the
    # two-phase results either way, and the same partial sums of squares
    for factor in (0.37, 1.0):
        scale = torch.tensor([factor], device="cuda")
        wb, wb2 = table.clone(), bias.clone()
        ops.fg_row_update(comp, fg, _lib.RX_SEG_SGD, wb, lr=lr, scale=scale, compact2=comp2, table2=wb2)
        ws, ws2 = table.clone(), bias.clone()
        undo = torch.full((fg.n, 2 * dim), float("nan"), device="cuda")
        undo2 = torch.full((fg.n, 2), float("nan"), device="cuda")
        sq_s = ops.fg_reduce(T, g, fg, _lib.RX_SEG_SGD_SPEC, ws, target2=ws2, lr=lr, want_sq=True, deep=deep, undo=undo, undo2=undo2)
        assert torch.equal(sq_s, sq)
        assert torch.equal(ws, w) and torch.equal(ws2, w2)                  # the plain step, as RX_SEG_SGD leaves it
        assert torch.equal(undo[end][:, :dim], comp[end]) and torch.equal(undo[end][:, dim:], table[r[end]])
        assert torch.equal(undo2[end][:, 0], comp2[end]) and torch.equal(undo2[end][:, 1], bias[r[end]])
        ops.fg_row_update(undo, fg, _lib.RX_SEG_SGD_SPEC, ws, lr=lr, scale=scale, compact2=undo2, table2=ws2)
        assert torch.equal(ws, wb) and torch.equal(ws2, wb2), factor
        assert torch.equal(ws[~touched], table[~touched])


def test_two_backward_passes_before_one_clipped_step():
    """Gradient accumulation over two batches with a clip: both pending row updates reach the table
    one after the other, equal to two sequential dense updates with the common clip factor."""
    from recsys_amd import synthetic
    from recsys_amd.models import fields
    from recsys_amd.optim import FusedOptimizer
    dims = [7, 1000, 13, 5000, 3, 211]
    torch.manual_seed(0)
    m = fields.FieldFM(dims, 16).cuda()
    w0, b0 = m.feature_embedding.weight.detach().clone(), m.bias_embedding.weight.detach().clone()
    batches = [synthetic.criteo_batch(dims, 700, seed=s, device="cuda") for s in (1, 2)]
    opt = FusedOptimizer(m, kind="sgd", lr=0.1, clip_norm=0.05)
    opt.zero_grad()
    for b in batches:
        m.loss(b).backward()
    opt.step()
    # restatement: dense gradients of both batches, the optimizer's norm (the two gradients counted
    # separately, as its partial sums do), both updates applied with that factor
    offs = torch.tensor([0] + dims[:-1], device="cuda").cumsum(0)
    grads = []
    for b in batches:
        wr, br = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        idx = b["sparse"] + offs[None, :]
        e = wr[idx]
        s = e.sum(1)
        logits = br[idx].sum(1) + 0.5 * (s * s - (e * e).sum(1)).sum(-1, keepdim=True)
        torch.nn.functional.binary_cross_entropy_with_logits(logits, b["label"]).backward()
        grads.append((wr.grad, br.grad))
    total = sum(float((gw.double() ** 2).sum() + (gb.double() ** 2).sum()) for gw, gb in grads) ** 0.5
    assert_close(torch.tensor(opt.last_total_norm.item()), torch.tensor(total), rtol=1e-5, what="norm over both passes")
    scale = min(1.0, 0.05 / (total + 1e-6))
    w_exp, b_exp = w0.double(), b0.double()
    for gw, gb in grads:
        w_exp = w_exp - 0.1 * scale * gw.double()
        b_exp = b_exp - 0.1 * scale * gb.double()
    assert_close(m.feature_embedding.weight.detach(), w_exp.float(), rtol=1e-5, what="rows after two pending updates")
    assert_close(m.bias_embedding.weight.detach(), b_exp.float(), rtol=1e-5, what="bias after two pending updates")


def test_reduce_agrees_with_the_sorted_path_on_hot_rows():
    """Runs of thousands of positions (tiny fields) cross many wavefront windows: compare with the
    position-ordered reduce over the fully sorted list (rx_segreduce_fm)."""
    from recsys_amd import _lib, ops, synthetic
    dims = [3, 11, 600, 90000]
    B, dim = 20000, 16
    torch.manual_seed(1)
    n_rows = sum(dims)
    table = torch.randn(n_rows, dim, device="cuda")
    sparse = synthetic.criteo_batch(dims, B, seed=5, device="cuda")["sparse"]
    T = torch.randn(B, dim, device="cuda")
    g = torch.randn(B, device="cuda") / B
    fg = _group(sparse, dims)
    grad, grad2 = torch.zeros_like(table), torch.zeros(n_rows, device="cuda")
    ops.fg_reduce(T, g, fg, _lib.RX_SEG_DENSE, grad, table=table, target2=grad2)
    F = len(dims) - 1
    base = torch.tensor(_bases(dims)[:F], device="cuda")
    si = ops.fm_sort_dedup(table, sparse[:, :F], base, sparse[:, F:], _bases(dims)[F])
    old, old2 = torch.zeros_like(table), torch.zeros(n_rows, device="cuda")
    ops.segreduce_fm(T, g, F, si, _lib.RX_SEG_DENSE, old, table=table, target2=old2)
    assert_close(grad, old, rtol=1e-5, what="grad vs sorted path")
    assert_close(grad2, old2, rtol=1e-5, what="bias grad vs sorted path")


@pytest.mark.parametrize("kind", ["sgd", "adagrad", "rowwise_adagrad"])
def test_clip_factor_formed_inside_the_row_update_equals_the_separate_launch_bitwise(kind, monkeypatch):
    """One field-grouped table and nothing else (the FM): FusedOptimizer.step lets every block of the
    row update add the partial sums of squares itself (rx_fg_row_update_clip).  Weights, optimizer
    state, clip factor and norm must equal the path with the one-block clip launch bit for bit."""
    from recsys_amd import optim, synthetic
    from recsys_amd.models import fields
    dims = [7, 1000, 13, 5000, 3, 211, 40000]
    batches = [synthetic.criteo_batch(dims, 3000, seed=s, device="cuda") for s in (1, 2, 3)]
    out = []
    for separate in (True, False):
        monkeypatch.setattr(optim, "_SEPARATE_CLIP_LAUNCH", separate)
        torch.manual_seed(0)
        m = fields.FieldFM(dims, 16).cuda()
        opt = optim.FusedOptimizer(m, kind=kind, lr=0.1, clip_norm=0.02)
        norms = []
        for b in batches:
            opt.zero_grad()
            m.loss(b).backward()
            opt.step()
            norms.append(opt.last_total_norm.clone())
        state = [r.state.clone() for r in opt.rules if r.state is not None]
        out.append(([p.detach().clone() for p in m.parameters()], state, norms))
    (wa, sa, na), (wb, sb_, nb) = out
    assert all(torch.equal(x, y) for x, y in zip(wa, wb))
    assert len(sa) == len(sb_) and all(torch.equal(x, y) for x, y in zip(sa, sb_))
    assert all(torch.equal(x, y) for x, y in zip(na, nb))
    assert float(na[0]) > 0.02          # the clip was active


@pytest.mark.parametrize("line_rows", [False, True, "sorted_path", "sorted_path_d128"])
@pytest.mark.parametrize("model", ["fm", "deepfm"])
def test_speculative_clipped_sgd_equals_the_two_phase_step_bitwise(model, line_rows, monkeypatch):
    """FusedOptimizer(kind="sgd", clip_norm=...) over field groups takes the step inside the reduce and only
    redoes rows when the norm exceeds the bound.  With a bound that bites on every step, on none, and on some
    (it sits between the norms of the steps), losses, norms and weights must equal the two-phase form
    (RX_NO_SPECULATIVE_SGD) bit for bit -- for the FM (clip factor formed inside the row update) and for the
    DeepFM, whose tower gradients share the norm (clip factor from its own launch)."""
    from recsys_amd import optim, synthetic
    from recsys_amd.models import fields
    if model == "deepfm" and line_rows is True:
        pytest.skip("the whole-line layout is an FM table option")
    D = 16
    if isinstance(line_rows, str):      # the general sort + run-walking reduce (rx_segreduce_fm) instead of field groups
        monkeypatch.setenv("RX_NO_FIELD_GROUPS", "1")
        monkeypatch.setattr(optim, "_SPECULATE_SORTED", True)     # off by default there (slower on wide rows)
        D = 128 if line_rows.endswith("d128") else 16
        line_rows = False
    dims = [7, 1000, 13, 5000, 3, 211, 40000]
    mk = (lambda s: synthetic.criteo_batch(dims, 3000, seed=s, device="cuda"))
    batches = [mk(s) for s in (1, 2, 3, 4)]

    def run(clip, speculate):
        monkeypatch.setattr(optim, "_NO_SPECULATION", not speculate)
        torch.manual_seed(0)
        with torch.device("cuda"):
            m = fields.FieldFM(dims, D, line_rows=line_rows) if model == "fm" else fields.FieldDeepFM(dims, D, n_dense=13, dropout=0.0)
            if D > 16:
                with torch.no_grad():
                    m.feature_embedding.weight.mul_(0.1)     # wide N(0, 1) rows saturate every logit
        m.train()
        opt = optim.FusedOptimizer(m, kind="sgd", lr=0.1, clip_norm=clip, speculative=True)   # (the default leaves models with a tower out)
        assert optim.FusedOptimizer(m, kind="sgd", lr=0.1, clip_norm=clip).speculative == (model == "fm")
        opt = optim.FusedOptimizer(m, kind="sgd", lr=0.1, clip_norm=clip, speculative=True)   # the rules of the last one made are the ones in use
        losses, norms = [], []
        for b in batches:
            opt.zero_grad()
            loss = m.loss(b)
            loss.backward()
            opt.step()
            losses.append(loss.detach().clone()); norms.append(opt.last_total_norm.clone())
        return [p.detach().clone() for p in m.parameters()], losses, norms

    _, _, norms = run(1e9, False)
    nv = sorted(float(x) for x in norms)
    assert nv[0] < nv[-1]
    for clip in (0.5 * nv[0], 2.0 * nv[-1], 0.5 * (nv[1] + nv[2])):
        wa, la, na = run(clip, False)
        wb, lb, nb = run(clip, True)
        assert all(torch.equal(x, y) for x, y in zip(la, lb)), clip
        assert all(torch.equal(x, y) for x, y in zip(na, nb)), clip
        assert all(torch.equal(x, y) for x, y in zip(wa, wb)), clip


@pytest.mark.parametrize("line_rows", [False, True, "sorted_path"])
def test_forward_between_backward_and_step_sees_the_rows_of_before_the_step(line_rows, monkeypatch):
    """The speculative clipped SGD step moves rows during the backward pass.  A forward pass that comes
    before step() (an evaluation, or the next micro-batch of a gradient accumulation) must still see the old
    rows: it puts them back, and step() then applies the records with the final factor -- same weights as
    with the two-phase form, bit for bit.  zero_grad() without a step drops the update altogether."""
    from recsys_amd import optim, synthetic
    from recsys_amd.models import fields
    if line_rows == "sorted_path":
        monkeypatch.setenv("RX_NO_FIELD_GROUPS", "1")
        monkeypatch.setattr(optim, "_SPECULATE_SORTED", True)
        line_rows = False
    dims = [7, 1000, 13, 5000, 3, 211]
    batch = synthetic.criteo_batch(dims, 900, seed=3, device="cuda")
    other = synthetic.criteo_batch(dims, 900, seed=4, device="cuda")

    def fresh():
        torch.manual_seed(0)
        with torch.device("cuda"):
            m = fields.FieldFM(dims, 16, line_rows=line_rows)
        return m.train()

    def two_phase(clip):
        m = fresh()
        old = optim._NO_SPECULATION
        optim._NO_SPECULATION = True
        try:
            opt = optim.FusedOptimizer(m, kind="sgd", lr=0.1, clip_norm=clip)
            opt.zero_grad(); m.loss(batch).backward(); opt.step()
        finally:
            optim._NO_SPECULATION = old
        return [p.detach().clone() for p in m.parameters()]

    for clip in (1e9, 1e-3):          # idle and biting
        ref = two_phase(clip)
        m = fresh()
        w0 = [p.detach().clone() for p in m.parameters()]
        logits0 = m(other).detach().clone()
        opt = optim.FusedOptimizer(m, kind="sgd", lr=0.1, clip_norm=clip)
        opt.zero_grad()
        m.loss(batch).backward()
        assert opt._has_spec and not torch.equal(m.feature_embedding.weight.detach(), w0[0])   # rows already moved
        assert torch.equal(m(other).detach(), logits0)                                          # ... and were put back for this
        assert all(torch.equal(p.detach(), q) for p, q in zip(m.parameters(), w0))
        opt.step()
        assert all(torch.equal(p.detach(), q) for p, q in zip(m.parameters(), ref)), clip
        # dropped gradients: backward, then zero_grad without a step
        m2 = fresh()
        opt2 = optim.FusedOptimizer(m2, kind="sgd", lr=0.1, clip_norm=clip)
        opt2.zero_grad(); m2.loss(batch).backward(); opt2.zero_grad()
        assert all(torch.equal(p.detach(), q) for p, q in zip(m2.parameters(), w0))


@pytest.mark.parametrize("kind,clip", [("sgd", 1.0), ("sgd", None), ("adagrad", 0.02)])
def test_loss_written_by_the_reduce_equals_the_forward_launch_bitwise(kind, clip, monkeypatch):
    """Inside a train step (utils.train_step, graph.GraphedTrainStep) the FM forward leaves its loss partials
    and the backward's reduce adds them up (one launch less between forward and reduce).  Same loss bits and
    same weights as a step whose forward finishes the loss itself; a plain model.loss() outside a step is
    complete as soon as it returns."""
    from recsys_amd import autograd as rxa, synthetic
    from recsys_amd.models import fields
    from recsys_amd.optim import FusedOptimizer
    from recsys_amd.utils import train_step
    from recsys_amd import ops
    dims = [7, 1000, 13, 5000, 3, 211]
    batches = [synthetic.criteo_batch(dims, 1100, seed=s, device="cuda") for s in (1, 2)]
    out = []
    taken = []          # did the forward leave the final sum out?
    orig = ops.fm2_fwd_bce
    monkeypatch.setattr(ops, "fm2_fwd_bce", lambda *a, **k: (taken.append(bool(k.get("defer_final"))), orig(*a, **k))[1])
    for deferred in (False, True):
        torch.manual_seed(0)
        with torch.device("cuda"):
            m = fields.FieldFM(dims, 16).train()
        opt = FusedOptimizer(m, kind=kind, lr=0.1, clip_norm=clip)
        losses = []
        for b in batches:
            if deferred:
                losses.append(train_step(m, opt, b, clip).clone())
            else:
                opt.clip_norm = clip
                opt.zero_grad()
                loss = m.loss(b)
                assert not rxa.DEFER_LOSS_FINAL[0]
                first = loss.detach().clone()               # complete before any backward pass
                loss.backward(); opt.step()
                assert torch.equal(first, loss.detach())
                losses.append(first)
        out.append((losses, [p.detach().clone() for p in m.parameters()]))
    assert taken == [False, False, True, True]
    (la, wa), (lb, wb) = out
    assert all(torch.equal(x, y) for x, y in zip(la, lb)), (la, lb)
    assert all(torch.equal(x, y) for x, y in zip(wa, wb))
    assert all(0.0 < float(x) < 100.0 for x in lb)


# ------------------------------------------------------------------ rows in whole cache lines
def _line_model(dims, dim, line_rows, how):
    from recsys_amd.models import fields
    torch.manual_seed(0)
    if how == "on_device":
        with torch.device("cuda"):
            return fields.FieldFM(dims, dim, line_rows=line_rows)
    return fields.FieldFM(dims, dim, line_rows=line_rows).cuda()      # rebuilt by .cuda(): must pack again


@pytest.mark.parametrize("how", ["on_device", "moved"])
def test_line_rows_keep_names_shapes_and_values(how):
    dims = [7, 1000, 13, 211]
    a, b = _line_model(dims, 16, False, how), _line_model(dims, 16, True, how)   # same generator, same draws
    assert b.lines_packed() and not a.lines_packed()
    sa, sb_ = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb_)
    for k in sa:
        assert sa[k].shape == sb_[k].shape and torch.equal(sa[k].cpu(), sb_[k].cpu())
    fw, bw = b.feature_embedding.weight, b.bias_embedding.weight
    assert fw.stride() == (32, 1) and bw.stride(0) == 32 and bw.data_ptr() == fw.data_ptr() + 64
    # a state_dict of the back-to-back model loads in place: the views stay views of one buffer
    with torch.no_grad():
        for p in a.parameters():
            p.add_(1.0)
    b.load_state_dict(a.state_dict())
    assert b.lines_packed() and torch.equal(b.bias_embedding.weight.cpu(), a.bias_embedding.weight.cpu())


@pytest.mark.parametrize("kind,clip", [("sgd", 0.02), ("sgd", None), ("adagrad", 0.02), ("adagrad", None), ("rowwise_adagrad", 0.02)])
def test_line_rows_train_bit_identically_to_back_to_back_rows(kind, clip):
    """Same seed, same batches: a model whose rows sit in whole 128-byte lines (bias entry behind the
    row) must produce the same loss bits and the same weights as the back-to-back layout after three
    fused steps, with the clip (two-phase update, clip factor inside the row update) and without
    (update inside the reduce); the padding behind the bias entry stays zero."""
    from recsys_amd import synthetic
    from recsys_amd.optim import FusedOptimizer
    dims = [7, 1000, 13, 5000, 3, 211, 40000]
    batches = [synthetic.criteo_batch(dims, 3000, seed=s, device="cuda") for s in (1, 2, 3)]
    out = []
    for line_rows in (False, True):
        m = _line_model(dims, 16, line_rows, "on_device")
        m.train()
        opt = FusedOptimizer(m, kind=kind, lr=0.1, clip_norm=clip)
        losses = []
        for b in batches:
            opt.zero_grad()
            loss = m.loss(b)
            loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
        m.eval()
        logits = m(batches[0]).detach().clone()
        out.append((m, losses, logits, [r.state.clone() for r in opt.rules if r.state is not None]))
    (ma, la, ea, sa), (mb, lb, eb, sb_) = out
    assert mb.lines_packed()
    assert all(torch.equal(x, y) for x, y in zip(la, lb))
    assert torch.equal(ea, eb)
    assert torch.equal(ma.feature_embedding.weight.detach(), mb.feature_embedding.weight.detach())
    assert torch.equal(ma.bias_embedding.weight.detach(), mb.bias_embedding.weight.detach())
    assert len(sa) == len(sb_) and all(torch.equal(x, y) for x, y in zip(sa, sb_))
    fw = mb.feature_embedding.weight
    whole = torch.as_strided(fw.detach(), (fw.shape[0], 32), (32, 1))
    # (the clipped SGD step keeps the bias gradient and the previous bias value in columns 17 and 18)
    assert int((whole[:, 19 if (kind, clip) == ("sgd", 0.02) else 17:] != 0).sum()) == 0
    assert not torch.equal(ma.feature_embedding.weight.detach(), _line_model(dims, 16, False, "on_device").feature_embedding.weight.detach())


def test_line_rows_dense_gradients_and_torch_optimizer():
    """Drop-in mode (no fused optimizer): dense gradients, clip_grad_norm_ and torch.optim.SGD on the
    strided parameters equal the back-to-back model's bit for bit."""
    from recsys_amd import synthetic
    dims = [7, 1000, 13, 5000, 3, 211]
    batch = synthetic.criteo_batch(dims, 2000, seed=4, device="cuda")
    res = []
    for line_rows in (False, True):
        m = _line_model(dims, 16, line_rows, "on_device")
        opt = torch.optim.SGD(m.parameters(), lr=0.1)
        for _ in range(2):
            opt.zero_grad()
            m.loss(batch).backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), 0.05)
            opt.step()
        res.append([p.detach().clone() for p in m.parameters()] + [p.grad.detach().clone() for p in m.parameters()])
    assert all(torch.equal(x, y) for x, y in zip(*res))


def test_line_rows_refuse_the_general_sort_path(monkeypatch):
    """The general sort + reduce takes back-to-back rows only: a whole-line model that is sent there
    must fail loudly, not scribble over memory."""
    from recsys_amd import autograd as rxa, synthetic
    from recsys_amd.optim import FusedOptimizer
    dims = [7, 1000, 13, 211]
    m = _line_model(dims, 16, True, "on_device")
    before = m.feature_embedding.weight.detach().clone()
    opt = FusedOptimizer(m, kind="sgd", lr=0.1, clip_norm=1.0)
    monkeypatch.setattr(rxa, "use_field_groups", lambda *a, **k: False)
    batch = synthetic.criteo_batch(dims, 500, seed=1, device="cuda")
    with pytest.raises((ValueError, RuntimeError)):
        opt.zero_grad()
        m.loss(batch).backward()
        opt.step()
    assert torch.equal(before, m.feature_embedding.weight.detach())
