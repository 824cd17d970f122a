"""Drop-in modules on the GPU against (a) the golden vectors captured from the imported
reference and (b) the CPU oracle on fresh seeded inputs at larger sizes.

Tolerance: rtol 1e-5 of the tensor's magnitude for fp32 logits, gradients and updated weights
(conftest.assert_close); the deduped row sets / untouched rows are exact."""
import inspect

import numpy as np
import pytest
import torch

from conftest import (MODEL_FIXTURES, adagrad_well_conditioned, assert_close, group, load_golden,
                      model_kwargs, to_torch)

pytestmark = pytest.mark.gpu


def _cls(name):
    from recsys_amd.models import matrix_factorization as mf, neuralcf as ncf
    return getattr(mf, name, None) or getattr(ncf, name)


def _build(rec, name, state=None):
    cls = _cls(name)
    kw = {k: v for k, v in model_kwargs(rec).items() if k in inspect.signature(cls.__init__).parameters}
    m = cls(**kw)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.load_state_dict(to_torch(group(rec, "state") if state is None else state))
    return m.cuda()


@pytest.mark.parametrize("fixture", MODEL_FIXTURES)
def test_golden_forward_loss_grads(fixture):
    rec = load_golden(fixture)
    name = fixture.split("_d")[0]
    batch = to_torch(group(rec, "batch"), "cuda")
    m = _build(rec, name).eval()
    with torch.no_grad():
        assert_close(m(batch).cpu().numpy(), rec["eval_logits"], what="eval logits")
        if "pw_logits" in rec:
            pw = dict(batch, item_code=batch["item_code"][:, 0].contiguous())
            out = m(pw, pointwise=True)
            assert out.shape == rec["pw_logits"].shape
            assert_close(out.cpu().numpy(), rec["pw_logits"], what="pointwise logits")
    m = _build(rec, name).train()
    assert_close(m(batch).detach().cpu().numpy(), rec["train_logits"], what="train logits")
    m = _build(rec, name).train()
    loss = m.loss(batch)
    assert loss.dtype == torch.float64
    assert_close(loss.item(), rec["loss"], rtol=1e-6, what="loss")
    loss.backward()
    params = dict(m.named_parameters())
    for k, g in group(rec, "grad").items():
        assert params[k].grad is not None, k
        assert_close(params[k].grad.cpu().numpy(), g, what=f"grad {k}", atol=1e-7)
        if g.ndim == 2 and k.endswith("embedding.weight"):
            untouched = ~np.abs(g).any(axis=1)
            assert not params[k].grad.cpu().numpy()[untouched].any(), f"{k}: untouched rows must be exactly 0"
    total = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    assert_close(total.item(), rec["clip_total_norm"], what="clip norm")


@pytest.mark.parametrize("fixture", MODEL_FIXTURES)
@pytest.mark.parametrize("opt", ["sgd", "adagrad", "adam"])
def test_golden_steps_with_torch_optimizers(fixture, opt):
    """Drop-in mode: the reference train step (recsys/utils.py:96-103) with torch's own optimizers."""
    rec = load_golden(fixture)
    name = fixture.split("_d")[0]
    batch = to_torch(group(rec, "batch"), "cuda")
    m = _build(rec, name).train()
    o = {"sgd": lambda p: torch.optim.SGD(p, lr=0.1), "adagrad": lambda p: torch.optim.Adagrad(p, lr=0.1),
         "adam": lambda p: torch.optim.Adam(p, lr=1e-3)}[opt](m.parameters())
    grads = group(rec, "grad")
    gmax = max(float(np.abs(g).max()) for g in grads.values())
    for s in (1, 2, 3):
        o.zero_grad()
        m.loss(batch).backward()
        if opt == "adam":
            torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        o.step()
        if s in (1, 3):
            _compare_state(m, group(rec, f"{opt}{s}"), opt, s, grads, gmax)


def _compare_state(m, expected, opt, s, grads, gmax):
    sd = m.state_dict()
    for k, v in expected.items():
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v)
            continue
        if k.endswith("running_mean") and opt != "sgd":
            continue
        a = sd[k].cpu().numpy()
        if opt in ("adagrad", "adam") and k in grads:
            a, v = adagrad_well_conditioned(a, v, grads[k], gmax)
        rtol = 1e-5 if s == 1 else (2e-5 if opt == "sgd" else 1e-4)
        assert_close(a, v, rtol=rtol, what=f"{opt}{s} {k}")


@pytest.mark.parametrize("fixture", MODEL_FIXTURES)
@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
def test_golden_steps_with_fused_optimizer(fixture, opt):
    """Fused sparse row update (no dense gradient for the tables) reproduces torch's dense
    SGD / Adagrad on the reference module."""
    from recsys_amd.optim import FusedOptimizer
    rec = load_golden(fixture)
    name = fixture.split("_d")[0]
    batch = to_torch(group(rec, "batch"), "cuda")
    m = _build(rec, name).train()
    o = FusedOptimizer(m, kind=opt, lr=0.1, eps=1e-10)
    grads = group(rec, "grad")
    gmax = max(float(np.abs(g).max()) for g in grads.values())
    for s in (1, 2, 3):
        o.zero_grad()
        m.loss(batch).backward()
        for p in m.sparse_parameters():
            assert p.grad is None
        o.step()
        if s in (1, 3):
            _compare_state(m, group(rec, f"{opt}{s}"), opt, s, grads, gmax)


# ------------------------------------------------------------------ oracle at larger sizes
def _oracle_case(name, dim, B, N, seed, n_users=6040, n_items=3706, n_occ=21, max_ts=2316):
    from oracle.models import OracleModel, init_state
    state = init_state(name, n_users=n_users, n_items=n_items, embedding_dim=dim, n_occupations=n_occ,
                       max_timestamp_rank=max_ts, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    zipf = lambda n, hi: (torch.rand(n, generator=g) ** 3 * hi).long().clamp_(0, hi - 1)  # noqa: E731
    batch = {"user_code": zipf(B, n_users), "item_code": zipf(B * N, n_items).reshape(B, N),
             "user_occupation": torch.randint(0, n_occ, (B,), generator=g),
             "item_timestamp_rank": torch.randint(0, max_ts, (B,), generator=g),
             "label": torch.zeros(B, N, dtype=torch.float64)}
    batch["label"][:, 0] = 1
    oracle = OracleModel(name, state, n_users=n_users, n_items=n_items, n_occupations=n_occ, dropout=0.0)
    cls = _cls(name)
    kw = dict(n_users=n_users, n_items=n_items, n_occupations=n_occ, max_timestamp_rank=max_ts, embedding_dim=dim)
    kw = {k: v for k, v in kw.items() if k in inspect.signature(cls.__init__).parameters}
    m = cls(**kw)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.load_state_dict(state)
    return oracle, m.cuda(), batch


@pytest.mark.parametrize("name,dim,B,N", [
    ("VanillaMF", 64, 1024, 5), ("BiasMF", 32, 512, 5), ("SideFeaturesMF", 64, 512, 5), ("TemporalMF", 64, 777, 5),
    ("FactorizationMachine", 16, 2048, 5), ("FactorizationMachine", 128, 300, 1), ("GMF", 64, 1024, 5),
    ("MLP", 64, 1024, 5), ("NeuMF", 64, 2048, 5), ("NeuMF", 64, 8192, 5), ("WidenDeep", 16, 1024, 5),
    ("DeepFactorizationMachine", 16, 1024, 5), ("VanillaMF", 20, 100, 100), ("NeuMF", 8, 33, 100),
])
def test_oracle_parity_ml1m_shapes(name, dim, B, N):
    oracle, m, batch = _oracle_case(name, dim, B, N, seed=7)
    dbatch = {k: v.cuda() for k, v in batch.items()}
    oracle.train(); m.train()
    lo = oracle.loss(batch)
    lm = m.loss(dbatch)
    assert_close(lm.item(), lo.item(), rtol=1e-6, what="loss")
    lo.backward(); lm.backward()
    params = dict(m.named_parameters())
    for k, p in oracle.params.items():
        assert_close(params[k].grad.cpu().numpy(), p.grad.numpy(), what=f"grad {k}", atol=1e-7)
    oracle.eval(); m.eval()
    with torch.no_grad():
        assert_close(m(dbatch).cpu().numpy(), oracle(batch).numpy(), what="eval logits")


def test_out_of_range_index_raises():
    from recsys_amd.models.matrix_factorization import VanillaMF
    m = VanillaMF(10, 10, 16).cuda()
    bad = {"user_code": torch.tensor([0, 10]).cuda(), "item_code": torch.zeros(2, 5, dtype=torch.int64).cuda()}
    with pytest.raises(IndexError):
        m(bad)
    bad = {"user_code": torch.tensor([0, 1]).cuda(), "item_code": torch.full((2, 5), -1, dtype=torch.int64).cuda()}
    with pytest.raises(IndexError):
        m(bad)
    with pytest.raises(TypeError):
        m({"user_code": torch.tensor([0, 1], dtype=torch.int32).cuda(), "item_code": torch.zeros(2, 5, dtype=torch.int64).cuda()})
    with pytest.raises(ValueError):
        m({"user_code": torch.tensor([0, 1]), "item_code": torch.zeros(2, 5, dtype=torch.int64)})


def test_train_step_is_bitwise_reproducible():
    oracle, m, batch = _oracle_case("FactorizationMachine", 16, 4096, 5, seed=3)
    dbatch = {k: v.cuda() for k, v in batch.items()}
    outs = []
    for _ in range(2):
        m.zero_grad()
        m.loss(dbatch).backward()
        outs.append([p.grad.clone() for p in m.parameters()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


# ------------------------------------------------------------------ dense Adam (rx_dense_adam / rx_dense_adam_dstep)
@pytest.mark.parametrize("fixture", ["NeuMF_d16", "DeepFactorizationMachine_d16", "VanillaMF_d64"])
def test_dense_adam_kernel_reproduces_reference_first_step(fixture):
    """The reference's train step with torch.optim.Adam(lr=1e-3) after clip_grad_norm_(1.0)
    (recsys/utils.py:96-103, recsys/pipelines/neuralcf/neumf.py:10): the golden capture holds every
    parameter before the step, its gradient, the clip norm and the parameter after step 1, so the
    kernel is fed exactly those and must land on the captured weights."""
    from recsys_amd import ops
    rec = load_golden(fixture)
    state, grads, after = group(rec, "state"), group(rec, "grad"), group(rec, "adam1")
    gmax = max(float(np.abs(g).max()) for g in grads.values())
    scale = torch.tensor([min(1.0, 1.0 / (float(rec["clip_total_norm"]) + 1e-6))], dtype=torch.float32, device="cuda")
    step_dev = torch.ones(1, dtype=torch.int32, device="cuda")
    for k, g in grads.items():
        for step in (1, step_dev):
            w = torch.from_numpy(state[k]).cuda().contiguous()
            m, v = torch.zeros_like(w), torch.zeros_like(w)
            ops.dense_adam(w, torch.from_numpy(g).cuda().contiguous(), m, v, 1e-3, 0.9, 0.999, 1e-8, step, scale)
            a, e = adagrad_well_conditioned(w.cpu().numpy(), after[k], g, gmax)
            assert_close(a, e, rtol=1e-5, what=f"adam1 {k}")


def test_dense_adam_steps_track_torch_and_the_device_counter_matches_the_host_one():
    from recsys_amd import ops
    torch.manual_seed(4)
    w0 = torch.randn(100_003, device="cuda")
    gs = [torch.randn_like(w0) * (0.5 + s) for s in range(4)]
    ref = w0.clone().requires_grad_(True)
    o = torch.optim.Adam([ref], lr=1e-2, betas=(0.8, 0.99), eps=1e-7)
    wh, mh, vh = w0.clone(), torch.zeros_like(w0), torch.zeros_like(w0)
    wd, md, vd = w0.clone(), torch.zeros_like(w0), torch.zeros_like(w0)
    counter = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.dense_adam(wd, gs[0], md, vd, 1e-2, 0.8, 0.99, 1e-7, counter)          # counter 0: no step taken
    assert torch.equal(wd, w0) and not md.any() and not vd.any()
    for s, g in enumerate(gs, 1):
        ref.grad = g.clone()
        o.step()
        ops.dense_adam(wh, g, mh, vh, 1e-2, 0.8, 0.99, 1e-7, s)
        counter.add_(1)
        ops.dense_adam(wd, g, md, vd, 1e-2, 0.8, 0.99, 1e-7, counter)
        assert torch.equal(wh, wd) and torch.equal(mh, md) and torch.equal(vh, vd)
        assert_close(wh.cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, what=f"adam step {s}")


def test_fused_optimizer_with_dense_adam_reproduces_reference_first_step():
    """FusedOptimizer(dense_kind="adam"): tables by fused sparse SGD, tower by rx_dense_adam with the
    global clip factor over all gradients -- the tower weights after step 1 are the golden capture's."""
    from recsys_amd.optim import FusedOptimizer
    rec = load_golden("NeuMF_d16")
    batch = to_torch(group(rec, "batch"), "cuda")
    m = _build(rec, "NeuMF").train()
    o = FusedOptimizer(m, kind="sgd", lr=1e-3, clip_norm=1.0, dense_kind="adam")
    o.zero_grad(); m.loss(batch).backward(); o.step()
    assert_close(o.last_total_norm.item(), rec["clip_total_norm"], what="clip norm")
    grads, after = group(rec, "grad"), group(rec, "adam1")
    gmax = max(float(np.abs(g).max()) for g in grads.values())
    sparse = {id(p) for p in m.sparse_parameters()}
    sd = m.state_dict()
    checked = 0
    for k, p in m.named_parameters():
        if id(p) in sparse:
            continue
        a, e = adagrad_well_conditioned(sd[k].cpu().numpy(), after[k], grads[k], gmax)
        assert_close(a, e, rtol=1e-5, what=f"adam1 {k}")
        checked += 1
    assert checked >= 6


@pytest.mark.parametrize("fixture", MODEL_FIXTURES)
def test_reference_optimizer_as_a_graph_replay_hits_the_golden_adam_steps(fixture):
    """The reference's optimizer -- Adam(lr=1e-3) over all parameters behind clip_grad_norm_(1.0)
    (recsys/pipelines/neuralcf/neumf.py:10, recsys/utils.py:101-103) -- through the graph path:
    FusedOptimizer.from_torch(torch.optim.Adam(...)) gives kind="dense_adam", the step is recorded WITHOUT a
    warm-up step (the device-side step count must start at zero and advance with the replays) and replayed
    three times.  Weights after replay 1 and 3 are the golden adam1 / adam3 captures, and bit for bit those
    of three eager steps of the same optimizer."""
    from recsys_amd import ops
    from recsys_amd.graph import GraphedTrainStep
    from recsys_amd.optim import FusedOptimizer
    rec = load_golden(fixture)
    name = fixture.split("_d")[0]
    batch = to_torch(group(rec, "batch"), "cuda")
    grads = group(rec, "grad")
    gmax = max(float(np.abs(g).max()) for g in grads.values())

    def make():
        m = _build(rec, name).train()
        o = FusedOptimizer.from_torch(torch.optim.Adam(m.parameters(), lr=1e-3), m, clip_norm=1.0)
        assert o is not None and o.kind == "dense_adam" and not o.sparse and o.clip_norm == 1.0
        return m, o

    m, o = make()
    eager = []
    for s in (1, 2, 3):
        o.zero_grad(); m.loss(batch).backward(); o.step()
        if s == 1:
            assert_close(o.last_total_norm.item(), rec["clip_total_norm"], what="clip norm")
        eager.append({k: v.detach().clone() for k, v in m.state_dict().items()})
    assert o.step_count == 3
    _compare_state(m, group(rec, "adam3"), "adam", 3, grads, gmax)

    m, o = make()
    ops.set_index_check("deferred")
    try:
        gs = GraphedTrainStep(m, o, warmup=0)
        slot = gs.add_batch({k: v.clone() for k, v in batch.items()})      # records, executes nothing
        assert o.step_count == 0
        for s in (1, 2, 3):
            gs(slot)
            if s == 1:
                _compare_state(m, group(rec, "adam1"), "adam", 1, grads, gmax)
            for k, v in m.state_dict().items():
                if not k.endswith("num_batches_tracked"):      # (BatchNorm's counter is a host-side add in the recording)
                    assert torch.equal(v, eager[s - 1][k]), (s, k)
        ops.raise_if_index_error()
    finally:
        ops.set_index_check("sync")
    assert o.step_count == 3


@pytest.mark.parametrize("B,N", [(8192, 5), (333, 5), (70, 1), (9, 64), (1, 7), (40, 13), (2049, 6), (515, 32), (77, 3)])
def test_neumf_one_launch_step_matches_the_per_layer_path(B, N, monkeypatch):
    """rx_neumf_step (forward + loss + backward of the list-wise NeuMF in one launch, embedding_dim 64) against
    the per-layer path (gathers, GMF product, MFMA Linear layers, rx_bce_fwd, their backward kernels): loss to
    fp64 rounding, every gradient to 1e-5 of its tensor's magnitude; tiles of whole samples are exercised with
    N that does and does not divide 64, a last partial tile, a single sample."""
    from recsys_amd import synthetic
    from recsys_amd.models.neuralcf import NeuMF
    d = synthetic.ML1M
    g = torch.Generator().manual_seed(B * 100 + N)
    batch = {"user_code": torch.randint(0, d["n_users"], (B,), generator=g).cuda(),
             "item_code": torch.randint(0, d["n_items"], (B, N), generator=g).cuda(),
             "label": (torch.rand(B, N, generator=g) < 0.3).double().cuda()}
    if B in (515, 77):
        batch["label"] = batch["label"].float()         # fp32 labels: the other instantiation of the kernel
    res = []
    for fused in (False, True):
        if fused:
            monkeypatch.delenv("RX_NO_FUSED_NEUMF", raising=False)
        else:
            monkeypatch.setenv("RX_NO_FUSED_NEUMF", "1")
        torch.manual_seed(3)
        m = NeuMF(d["n_users"], d["n_items"], 64, dropout=0.0).cuda().train()
        with torch.no_grad():        # N(0, 0.01) embeddings leave tiny gradients: make the tower see real signal
            for e in (m.user_embedding_gmf, m.item_embedding_gmf, m.user_embedding_mlp, m.item_embedding_mlp):
                e.weight.mul_(30.0)
            for lin in (m.linear_mlp[0], m.linear_mlp[3], m.linear_gmf, m.linear_final):
                lin.bias.normal_(0, 0.1)
        loss = m.loss(batch)
        assert (type(loss.grad_fn).__name__ == "NeuMFLossBackward") == fused
        loss.backward()
        res.append((loss.item(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}))
    (la, ga), (lb, gb) = res
    assert_close(lb, la, rtol=1e-6, what="loss")
    # A first-layer unit whose pre-activation is zero to fp32 rounding passes its gradient under one summation
    # order and not under another (ReLU's gate is h > 0 in both paths): where fp64 puts |z| below 1e-6 the rows
    # that unit feeds -- the pair's two tower rows, the unit's row of W1 and its bias -- are left out.
    with torch.no_grad():
        u = batch["user_code"][:, None].expand(B, N).reshape(-1)
        it = batch["item_code"].reshape(-1)
        x = torch.cat([m.user_embedding_mlp.weight[u], m.item_embedding_mlp.weight[it]], 1).double()
        z = x @ m.linear_mlp[0].weight.double().t() + m.linear_mlp[0].bias.double()
        near = (z.abs() < 1e-6).nonzero().cpu()
    assert len(near) <= 32
    skip = {"user_embedding_mlp.weight": u[near[:, 0]].cpu(), "item_embedding_mlp.weight": it[near[:, 0]].cpu(),
            "linear_mlp.0.weight": near[:, 1], "linear_mlp.0.bias": near[:, 1]}
    for k in ga:
        a, e = gb[k].cpu().clone(), ga[k].cpu().clone()
        if k in skip and len(near):
            a[skip[k]] = 0
            e[skip[k]] = 0
        assert_close(a.numpy(), e.numpy(), what=f"grad {k}", atol=1e-9)


def test_neumf_one_launch_step_dropout_stream():
    """The Dropout between the tower's layers inside rx_neumf_step: a counter-based draw keyed by the model's
    device counter.  The same counter value reproduces the step bit for bit, the launch advances it, the next
    call draws anew, and with p = 0.1 about a tenth of the first layer's units are off (seen through the
    gradient of its bias: a dropped unit passes no gradient)."""
    from recsys_amd import synthetic
    from recsys_amd.models.neuralcf import NeuMF
    d = synthetic.ML1M
    batch = synthetic.ml1m_batch(2048, seed=1, device="cuda")
    torch.manual_seed(0)
    m = NeuMF(d["n_users"], d["n_items"], 64, dropout=0.1).cuda().train()

    def step():
        m.zero_grad()
        loss = m.loss(batch)
        loss.backward()
        return loss.item(), m.linear_mlp[0].weight.grad.clone()

    l1, g1 = step()
    assert int(m._drop_seed.item()) == torch.initial_seed() % (2 ** 62) + 1
    l2, g2 = step()
    assert l1 != l2 and not torch.equal(g1, g2)
    m._drop_seed.sub_(2)
    l3, g3 = step()
    assert l3 == l1 and torch.equal(g3, g1)
    m.eval()
    with torch.no_grad():
        ev = m(batch)
    assert torch.isfinite(ev).all()


@pytest.mark.parametrize("name", ["VanillaMF", "BiasMF", "SideFeaturesMF", "TemporalMF", "GMF"])
@pytest.mark.parametrize("label_dtype", [torch.float64, torch.float32])
def test_fused_forward_loss_of_the_mf_family_equals_the_two_launch_path(name, label_dtype, monkeypatch):
    """rx_mf_score_fwd_bce (logits, loss and dL/dlogit from the forward launch) against rx_mf_score_fwd +
    rx_bce_fwd: the same dL/dlogit bits, hence bit-identical gradients and weights after fused steps; the
    loss agrees to fp64 rounding (its block partials are grouped differently)."""
    from recsys_amd import synthetic
    from recsys_amd.models import matrix_factorization as mf
    from recsys_amd.models.matrix_factorization import mf as mf_mod
    from recsys_amd.models.neuralcf import GMF
    from recsys_amd.optim import FusedOptimizer
    batches = [synthetic.ml1m_batch(300, seed=s, device="cuda") for s in (1, 2)]
    for b in batches:
        b["label"] = b["label"].to(label_dtype)
    res = []
    for fused in (False, True):
        monkeypatch.setattr(mf_mod, "_FUSED_LOSS", fused)
        if fused:
            monkeypatch.delenv("RX_NO_FUSED_MF_LOSS", raising=False)
        else:
            monkeypatch.setenv("RX_NO_FUSED_MF_LOSS", "1")
        torch.manual_seed(0)
        kw = dict(n_users=synthetic.ML1M["n_users"], n_items=synthetic.ML1M["n_items"], embedding_dim=16)
        if name in ("SideFeaturesMF", "TemporalMF"):
            kw["n_occupations"] = synthetic.ML1M["n_occupations"]
        if name == "TemporalMF":
            kw["max_timestamp_rank"] = synthetic.ML1M["max_timestamp_rank"]
        m = (GMF if name == "GMF" else getattr(mf, name))(**kw).cuda().train()
        opt = FusedOptimizer(m, kind="adagrad", lr=0.05, clip_norm=1.0)
        losses = []
        for b in batches:
            opt.zero_grad()
            loss = m.loss(b)
            if label_dtype == torch.float64:
                assert (type(loss.grad_fn).__name__ == "MFLossBackward") == fused
            loss.backward()
            opt.step()
            losses.append(float(loss))
        res.append((losses, {k: v.detach().clone() for k, v in m.state_dict().items()}))
    (la, wa), (lb, wb) = res
    for x, y in zip(la, lb):
        assert abs(x - y) <= 1e-6 * abs(x) if label_dtype == torch.float32 else abs(x - y) <= 1e-12 * abs(x)
    for k in wa:
        assert torch.equal(wa[k], wb[k]), k


@pytest.mark.parametrize("kind", ["sgd", "adagrad", "rowwise_adagrad"])
@pytest.mark.parametrize("fixture", ["VanillaMF_d16", "NeuMF_d16", "NeuMF_d64", "GMF_d16", "MLP_d16", "BiasMF_d16"])
def test_step_tail_in_one_launch_equals_the_three_launches_bitwise(fixture, kind, monkeypatch):
    """rx_step_tail (clip factor + row updates of all tables + dense update in one launch) against
    rx_clip_scale_parts + rx_row_update_multi + rx_multi_dense_update: weights, optimizer state and clip norm bit
    for bit after three clipped steps, for every ML-1M-sized model family and all three fused rules; the bound
    (0.05) bites on every step."""
    from recsys_amd import optim
    if fixture not in MODEL_FIXTURES:
        pytest.skip(f"no golden capture named {fixture}")
    rec = load_golden(fixture)
    batch = to_torch(group(rec, "batch"), "cuda")

    def run(fused):
        monkeypatch.setattr(optim, "_NO_FUSED_TAIL", not fused)
        calls = []
        orig = optim.ops.step_tail
        monkeypatch.setattr(optim.ops, "step_tail", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        m = _build(rec, fixture.split("_")[0]).train()
        o = optim.FusedOptimizer(m, kind=kind, lr=0.05, clip_norm=0.05)
        norms = []
        for _ in range(3):
            o.zero_grad(); m.loss(batch).backward(); o.step()
            norms.append(o.last_total_norm.clone())
        monkeypatch.setattr(optim.ops, "step_tail", orig)
        state = [st.clone() for r, p in zip(o.rules, o.sparse) for st in [r._state_for(p)] if st is not None]
        return [p.detach().clone() for p in m.parameters()], norms, state, len(calls)

    w1, n1, s1, c1 = run(True)
    w0, n0, s0, c0 = run(False)
    # the one-launch tail is what ran, and what did not (BiasMF's tables have two widths -- rows of 16 floats and of
    # one -- which is two update launches: it keeps the three-launch form)
    assert c0 == 0 and c1 == (0 if fixture.startswith("BiasMF") else 3)
    for a, b in zip(w1 + n1 + s1, w0 + n0 + s0):
        assert torch.equal(a, b)


@pytest.mark.parametrize("fixture", ["VanillaMF_d16", "BiasMF_d16", "GMF_d16", "SideFeaturesMF_d16", "TemporalMF_d16"])
def test_mf_loss_written_by_the_backward_launch_equals_the_forward_launch_bitwise(fixture, monkeypatch):
    """Inside a train step (utils.train_step, the graph recorders) rx_mf_score_fwd_bce leaves its loss partials and
    rx_mf_score_bwd adds them up (one launch less per step).  Same loss bits and same weights as steps whose forward
    finishes the loss itself; a plain model.loss() outside a step is complete as soon as it returns."""
    from recsys_amd import optim, ops
    from recsys_amd.utils import train_step
    if fixture not in MODEL_FIXTURES:
        pytest.skip(f"no golden capture named {fixture}")
    rec = load_golden(fixture)
    batch = to_torch(group(rec, "batch"), "cuda")
    out = []
    for deferred in (True, False):
        m = _build(rec, fixture.split("_")[0]).train()
        o = optim.FusedOptimizer(m, kind="sgd", lr=0.05, clip_norm=1.0)
        seen = []
        orig = ops.mf_score_bwd
        monkeypatch.setattr(ops, "mf_score_bwd", lambda *a, **k: (seen.append(k.get("loss_final") is not None), orig(*a, **k))[1])
        losses = []
        for _ in range(3):
            if deferred:
                losses.append(train_step(m, o, batch, 1.0).clone())
            else:
                o.zero_grad(); loss = m.loss(batch); loss.backward(); o.step()
                losses.append(loss.detach().clone())
        monkeypatch.setattr(ops, "mf_score_bwd", orig)
        assert seen == [deferred] * 3
        out.append((losses, [p.detach().clone() for p in m.parameters()]))
        plain = m.loss(batch)               # outside a step: written by the forward's own finishing launch
        assert torch.isfinite(plain).item()
    (la, wa), (lb, wb) = out
    assert all(torch.equal(x, y) for x, y in zip(la, lb))
    assert all(torch.equal(x, y) for x, y in zip(wa, wb))
