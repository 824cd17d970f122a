"""Size-independent properties at the full benchmark sizes (Criteo shape: batch 65536, 26 fields,
33 M rows, D = 16), where the CPU oracle would take minutes: sortedness and permutation checks
of the index preparation, linearity and run-to-run bit-reproducibility of the reduce, untouched
rows, and one fused train step against a dense-gradient restatement built from plain torch ops on
the same device."""
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu

B, D = 65536, 16


@pytest.fixture(scope="module")
def setup():
    from recsys_amd import ops, synthetic
    ops.set_index_check("sync")
    dev = torch.device("cuda", 0)
    dims = synthetic.criteo_field_dims()
    batch = synthetic.criteo_batch(dims, B, seed=7, device=dev)
    offs = torch.tensor([0] + dims[:-1], device=dev).cumsum(0)
    flat = (batch["sparse"] + offs[None, :]).reshape(-1).contiguous()
    return dict(ops=ops, dev=dev, dims=dims, batch=batch, flat=flat, rows=sum(dims))


def test_sort_dedup_properties_full_size(setup):
    ops, flat, rows = setup["ops"], setup["flat"], setup["rows"]
    si = ops.sort_dedup(flat, rows, want_inverse=True)
    nu = int(si.n_uniq.item())
    uniq = si.uniq_rows[:nu]
    assert torch.all(uniq[1:] > uniq[:-1]), "distinct rows must be strictly increasing"
    assert torch.equal(uniq, torch.unique(flat)), "distinct-row set"
    pos = si.sorted_pos.long()
    assert torch.equal(torch.sort(pos).values, torch.arange(flat.numel(), device=flat.device)), "a permutation"
    keys = flat[pos]
    assert torch.all(keys[1:] >= keys[:-1]), "sorted by row"
    same = keys[1:] == keys[:-1]
    assert torch.all(pos[1:][same] > pos[:-1][same]), "stable: original order inside a run"
    assert torch.equal(pos, torch.sort(flat, stable=True).indices), "the permutation of a stable sort"
    seg = si.seg_off[:nu + 1].long()
    assert seg[0] == 0 and seg[-1] == flat.numel() and torch.all(seg[1:] > seg[:-1])
    assert torch.equal(uniq[si.sorted_slot.long()], keys), "slot of every sorted position"
    assert torch.equal(uniq[si.inverse.long()], flat), "inverse map"


def test_segreduce_linearity_and_reproducibility_full_size(setup):
    from recsys_amd import _lib
    ops, flat, rows, dev = setup["ops"], setup["flat"], setup["rows"], setup["dev"]
    n = flat.numel()
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(n, D, generator=g, device=dev)
    b = torch.randn(n, D, generator=g, device=dev)
    si = ops.sort_dedup(flat, rows)
    nu = int(si.n_uniq.item())

    def compact(occ):
        out = torch.empty(n, D, device=dev)
        ops.segreduce(occ, si, _lib.RX_SEG_COMPACT, out)
        return out[:nu]

    ca, cb, cab = compact(a), compact(b), compact(a + b)
    assert torch.equal(ca, compact(a)), "bitwise reproducible run to run"
    assert_close((ca + cb).cpu().numpy(), cab.cpu().numpy(), rtol=1e-5, what="linearity")
    # checksum of checksums: the column sums of all distinct-row gradients equal those of all occurrences
    assert_close(ca.double().sum(0).cpu().numpy(), a.double().sum(0).cpu().numpy(), rtol=1e-5, what="column checksum")
    # against an atomics-free restatement with torch ops
    ref = torch.zeros(nu, D, device=dev, dtype=torch.float64).index_add_(0, si.inverse.long() if si.inverse is not None
                                                                         else torch.searchsorted(si.uniq_rows[:nu], flat), a.double())
    assert_close(ca.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, what="per-row sums")


def test_fm_forward_permutation_invariance_full_size(setup):
    """The FM logit sums over fields: feeding the non-item fields in another order changes nothing
    but the rounding."""
    from recsys_amd.models import fields
    dims, batch, dev = setup["dims"], setup["batch"], setup["dev"]
    torch.manual_seed(0)
    with torch.device(dev):
        m = fields.FieldFM(dims, D).eval()
    logits = m(batch)
    assert logits.shape == (B, 1) and torch.isfinite(logits).all()
    # same model evaluated by plain torch gathers on the device
    offs = torch.tensor([0] + dims[:-1], device=dev).cumsum(0)
    idx = batch["sparse"] + offs[None, :]
    e = m.feature_embedding.weight[idx].double()
    bias = m.bias_embedding.weight[idx].double().sum(1)
    s = e.sum(1)
    ref = bias + 0.5 * (s * s - (e * e).sum(1)).sum(-1, keepdim=True)
    assert_close(logits.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, what="FM logits at full size")


_FUSED_STEP_RESULT = {}


@pytest.mark.parametrize("line_rows", [False, True], ids=["rows_back_to_back", "rows_in_whole_lines"])
def test_fused_step_matches_dense_restatement_full_size(setup, line_rows):
    """One fused SGD step with global clip 1.0 at the benchmark size against dense gradients,
    clip_grad_norm_ and a dense SGD update made of plain torch ops; rows that the batch does not
    touch must stay bit-identical.  ``line_rows``: the layout bench.py times (every row in a 128-byte
    line of its own with the bias entry behind it) -- same parameters and arithmetic, so the two
    layouts must also agree with each other bit for bit (loss, norm, every weight)."""
    from recsys_amd.models import fields
    from recsys_amd.optim import FusedOptimizer
    dims, batch, dev = setup["dims"], setup["batch"], setup["dev"]
    torch.manual_seed(0)
    with torch.device(dev):
        m = fields.FieldFM(dims, D, line_rows=line_rows)
    assert m.lines_packed() == line_rows
    w0, b0 = m.feature_embedding.weight.detach().clone(), m.bias_embedding.weight.detach().clone()
    # reference: autograd through plain torch ops, dense gradients
    wr, br = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    offs = torch.tensor([0] + dims[:-1], device=dev).cumsum(0)
    idx = batch["sparse"] + offs[None, :]
    e = wr[idx]
    s = e.sum(1)
    logits = br[idx].sum(1) + 0.5 * (s * s - (e * e).sum(1)).sum(-1, keepdim=True)
    loss_ref = torch.nn.functional.binary_cross_entropy_with_logits(logits, batch["label"])
    loss_ref.backward()
    tn = torch.nn.utils.clip_grad_norm_([wr, br], 1.0)
    lr = 0.05
    with torch.no_grad():
        wr -= lr * wr.grad
        br -= lr * br.grad
    del e, s, logits
    # fused path
    opt = FusedOptimizer(m, kind="sgd", lr=lr, clip_norm=1.0)
    m.train()
    opt.zero_grad()
    loss = m.loss(batch)
    loss.backward()
    opt.step()
    assert_close(loss.item(), loss_ref.item(), rtol=1e-6, what="loss")  # fp64 mean over fp32 logits that agree to rounding
    assert_close(opt.last_total_norm.item(), tn.item(), rtol=1e-5, what="global gradient norm")
    touched = torch.zeros(sum(dims), dtype=torch.bool, device=dev)
    touched[idx.reshape(-1)] = True
    w1, b1 = m.feature_embedding.weight.detach(), m.bias_embedding.weight.detach()
    assert torch.equal(w1[~touched], w0[~touched]) and torch.equal(b1[~touched], b0[~touched]), "untouched rows moved"
    assert_close(w1[touched].cpu().numpy(), wr.detach()[touched].cpu().numpy(), rtol=1e-5, what="updated rows")
    assert_close(b1[touched].cpu().numpy(), br.detach()[touched].cpu().numpy(), rtol=1e-5, what="updated bias rows")
    if line_rows:   # the update writes the line whole: bias entry, its gradient and previous value, then zeros
        buf = w1.as_strided((sum(dims), 2 * D), (2 * D, 1), w1.storage_offset())
        assert not buf[:, D + 3:].any(), "padding columns of a line were written"
    mine = (loss.detach().clone(), opt.last_total_norm.clone(), w1[touched].contiguous().clone(), b1[touched].contiguous().clone())
    other = _FUSED_STEP_RESULT.get(not line_rows)
    _FUSED_STEP_RESULT[line_rows] = mine
    if other is not None:
        for a, b, what in zip(mine, other, ("loss", "norm", "rows", "bias rows")):
            assert torch.equal(a, b), f"{what}: the two table layouts disagree bitwise"
        _FUSED_STEP_RESULT.clear()


def test_fused_step_on_table_beyond_2G_elements():
    """A table whose element count passes 2^31 (18 M rows x 128 floats = 9.2 GB): rows at the very
    end of the table are gathered, reduced and updated at 64-bit offsets.  Reference: the same step
    on the compacted set of touched rows with plain torch ops (renumbering rows changes nothing)."""
    from recsys_amd import ops
    from recsys_amd.models import fields
    from recsys_amd.optim import FusedOptimizer
    ops.set_index_check("sync")
    dev = torch.device("cuda", 0)
    dims, Dw, Bs = [9_000_000, 50, 9_000_000], 128, 4096
    assert sum(dims) * Dw > 2**31
    g = torch.Generator().manual_seed(11)
    cols = []
    for d in dims:
        hi = torch.randint(max(d - 3000, 0), d, (Bs,), generator=g)        # the last rows of every field
        lo = torch.randint(0, d, (Bs,), generator=g)
        cols.append(torch.where(torch.rand(Bs, generator=g) < 0.7, hi, lo))
    sparse = torch.stack(cols, 1)
    sparse[0, 2] = dims[2] - 1                                              # the table's very last row
    batch = {"sparse": sparse.to(dev), "label": (torch.rand(Bs, 1, generator=g) < 0.25).double().to(dev)}
    torch.manual_seed(3)
    with torch.device(dev):
        m = fields.FieldFM(dims, Dw)
    w0, b0 = m.feature_embedding.weight.detach().clone(), m.bias_embedding.weight.detach().clone()
    offs = torch.tensor([0] + dims[:-1], device=dev).cumsum(0)
    idx = batch["sparse"] + offs[None, :]
    uniq, inv = torch.unique(idx, return_inverse=True)
    wc, bc = w0[uniq].clone().requires_grad_(True), b0[uniq].clone().requires_grad_(True)
    e = wc[inv]
    s = e.sum(1)
    logits = bc[inv].sum(1) + 0.5 * (s * s - (e * e).sum(1)).sum(-1, keepdim=True)
    loss_ref = torch.nn.functional.binary_cross_entropy_with_logits(logits, batch["label"])
    loss_ref.backward()
    tn = torch.nn.utils.clip_grad_norm_([wc, bc], 1.0)
    lr = 0.05
    with torch.no_grad():
        wc -= lr * wc.grad
        bc -= lr * bc.grad
    opt = FusedOptimizer(m, kind="sgd", lr=lr, clip_norm=1.0)
    m.train()
    opt.zero_grad()
    loss = m.loss(batch)
    loss.backward()
    opt.step()
    assert_close(loss.item(), loss_ref.item(), rtol=1e-6, what="loss")
    assert_close(opt.last_total_norm.item(), tn.item(), rtol=1e-5, what="global gradient norm")
    w1, b1 = m.feature_embedding.weight.detach(), m.bias_embedding.weight.detach()
    moved = (w1 != w0).any(1) | (b1 != b0).any(1)
    touched = torch.zeros(sum(dims), dtype=torch.bool, device=dev)
    touched[uniq] = True
    assert not (moved & ~touched).any(), "untouched rows moved"
    assert moved[sum(dims) - 1], "the last row was not updated"
    assert_close(w1[uniq].cpu().numpy(), wc.detach().cpu().numpy(), rtol=1e-5, what="updated rows")
    assert_close(b1[uniq].cpu().numpy(), bc.detach().cpu().numpy(), rtol=1e-5, what="updated bias rows")


def test_deepfm_step_matches_dense_restatement_full_size(setup):
    """FieldDeepFM at the benchmark size (batch 65536, 26 fields, 33 M rows, D = 16, 13 dense
    features, BatchNorm tower 432-432-216-108-1): eval logits, then one fused clip + SGD step, against
    the same model written with plain torch ops on the device (index gathers, nn.Sequential tower,
    dense table gradients, clip_grad_norm_, dense SGD).  Rows the batch does not touch stay
    bit-identical and a second run of the fused step reproduces the first bit for bit."""
    import copy
    from recsys_amd.models import fields
    from recsys_amd.optim import FusedOptimizer
    dims, batch, dev = setup["dims"], setup["batch"], setup["dev"]
    torch.manual_seed(0)
    with torch.device(dev):
        m = fields.FieldDeepFM(dims, D, n_dense=13, dropout=0.0)
    tower = copy.deepcopy(m.deep_mlp)
    w0, b0 = m.feature_embedding.weight.detach().clone(), m.bias_embedding.weight.detach().clone()
    start = copy.deepcopy(m.state_dict())
    offs = torch.tensor([0] + dims[:-1], device=dev).cumsum(0)
    idx = batch["sparse"] + offs[None, :]

    def restated(w, b, net):
        e = w[idx]
        s = e.sum(1)
        fm = b[idx].sum(1) + 0.5 * (s * s - (e * e).sum(1)).sum(-1, keepdim=True)
        x = torch.cat([e[:, -1], e[:, :-1].reshape(B, -1), batch["dense"]], 1)
        x = torch.nn.functional.pad(x, (0, net[0].in_features - x.shape[1]))
        return fm + net(x)

    m.eval(); tower.eval()
    with torch.no_grad():
        assert_close(m(batch).cpu().numpy(), restated(w0, b0, tower).cpu().numpy(), rtol=1e-5, what="DeepFM eval logits at full size")

    m.train(); tower.train()
    wr, br = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    loss_ref = torch.nn.functional.binary_cross_entropy_with_logits(restated(wr, br, tower), batch["label"])
    loss_ref.backward()
    ref_params = [wr, br] + list(tower.parameters())
    tn = torch.nn.utils.clip_grad_norm_(ref_params, 1.0)
    lr = 0.05
    with torch.no_grad():
        for p in ref_params:
            p -= lr * p.grad

    def fused_step():
        m.load_state_dict(start)
        opt = FusedOptimizer(m, kind="sgd", lr=lr, clip_norm=1.0)
        m.train()
        opt.zero_grad()
        loss = m.loss(batch)
        loss.backward()
        opt.step()
        return loss, opt.last_total_norm.item(), {k: v.detach().clone() for k, v in m.state_dict().items()}

    loss, norm, after = fused_step()
    assert_close(loss.item(), loss_ref.item(), rtol=1e-5, what="loss")
    assert_close(norm, tn.item(), rtol=1e-5, what="global gradient norm")
    touched = torch.zeros(sum(dims), dtype=torch.bool, device=dev)
    touched[idx.reshape(-1)] = True
    w1, b1 = after["feature_embedding.weight"], after["bias_embedding.weight"]
    assert torch.equal(w1[~touched], w0[~touched]) and torch.equal(b1[~touched], b0[~touched]), "untouched rows moved"
    assert_close(w1[touched].cpu().numpy(), wr.detach()[touched].cpu().numpy(), rtol=1e-5, what="updated rows")
    assert_close(b1[touched].cpu().numpy(), br.detach()[touched].cpu().numpy(), rtol=1e-5, what="updated bias rows")
    ref_tower = tower.state_dict()
    for k, v in ref_tower.items():
        if k.endswith("num_batches_tracked"):
            continue
        assert_close(after["deep_mlp." + k].cpu().numpy(), v.cpu().numpy(), rtol=2e-5, what="deep_mlp." + k, atol=1e-7)
    _, _, again = fused_step()
    for k, v in after.items():
        assert torch.equal(again[k], v), f"{k}: not reproducible run to run"
