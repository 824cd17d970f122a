"""End-to-end pieces around the hot path on the GPU: the train / eval loop mirror, device-side
ranking, the sharded table path at world size 1 with the real kernels."""
import numpy as np
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu


def _small_data():
    from recsys_amd.dataset import ImplicitData, synthetic_ml1m_data
    d = synthetic_ml1m_data(seed=3, n_users=400, n_items=300, n_occupations=7, n_train=12000, pool=60, n_test_codes=50)
    return ImplicitData(d, device="cuda", train_batch_size=256, test_batch_size=128)


@pytest.mark.parametrize("fused", [False, True, "graph", "graph_torch_adam"])
def test_train_loop_learns(fused, capsys):
    """recsys/utils.py:84-117 call pattern: eval before training, then epochs of train + eval."""
    from recsys_amd.models.neuralcf import NeuMF
    from recsys_amd.optim import FusedOptimizer
    from recsys_amd.utils import load_model, train_loop
    torch.manual_seed(0)
    data = _small_data()
    model = load_model(NeuMF, "cuda", n_users=data.n_users, n_items=data.n_items, embedding_dim=16)
    if fused and fused != "graph_torch_adam":
        opt = FusedOptimizer(model, kind="adagrad", lr=0.05, clip_norm=1.0)
    else:       # the reference pipelines' optimizer; with graph=True train_loop swaps in its fused equivalent
        opt = torch.optim.Adam(model.parameters(), lr=5e-3)
    hist = train_loop(model, data, opt, num_epochs=4, clip_norm=1.0, k=10, graph=str(fused).startswith("graph"))
    out = capsys.readouterr().out
    assert out.count("metrics - ndcg@10") == 5 and "epoch [2/4], loss:" in out
    losses = [h[0] for h in hist]
    assert losses[-1] < losses[0] < 0.75
    assert all(0.0 <= h[1] <= 1.0 and 0.0 <= h[2] <= 1.0 for h in hist)


def test_eval_ranking_on_device_matches_host():
    from recsys_amd.metrics import compute_metrics
    from recsys_amd.models.matrix_factorization import VanillaMF
    from recsys_amd.utils import topk
    torch.manual_seed(1)
    data = _small_data()
    data.create_test_dataloader()
    model = VanillaMF(data.n_users, data.n_items, 32).cuda().eval()
    with torch.no_grad():
        scores = torch.cat([model(b) for b in data.test_dataloader])
    assert scores.shape == (data.n_users, 50)
    dev = topk(scores, array=data.test_codes, k=10).cpu().numpy()
    host = topk(scores.cpu().numpy(), array=data.test_codes, k=10)
    assert np.array_equal(dev, host)
    a = compute_metrics(data.test_true, torch.from_numpy(dev).cuda())
    b = compute_metrics(data.test_true, host)
    assert a == b


def test_epoch_loader_on_device():
    data = _small_data()
    dl = data.create_negative_sampled_train_dataloader()
    n = 0
    for b in dl:
        assert all(v.is_cuda for v in b.values())
        assert b["item_code"].dtype == torch.int64 and b["label"].dtype == torch.float64
        n += b["user_code"].shape[0]
    assert n == data.implicit_matrix.nnz and len(dl) == -(-n // 256)


@pytest.mark.parametrize("fixed", [False, True])
@pytest.mark.parametrize("kind", ["sgd", "adagrad", None])
def test_sharded_fm_world1_matches_field_fm(kind, fixed):
    """World size 1: the exchange degenerates to local permutations, all through the HIP kernels."""
    from recsys_amd.models.fields import FieldFM
    from recsys_amd.optim import FusedOptimizer
    from recsys_amd.sharded import ShardedFieldFM
    dims = [7, 1000, 13, 5000, 3, 211]
    torch.manual_seed(2)
    ref = FieldFM(dims, 16).cuda()
    state = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    sh = ShardedFieldFM(dims, 16, "cuda", full_state=state)
    g = torch.Generator().manual_seed(4)
    batch = {"sparse": torch.stack([torch.randint(0, d, (333,), generator=g) for d in dims], 1).cuda(),
             "label": (torch.rand(333, 1, generator=g) < 0.3).double().cuda()}
    if fixed:
        sh.calibrate([batch])
    if kind:
        opt = FusedOptimizer(ref, kind=kind, lr=0.1)
        sh.set_rule(kind, 0.1)
    for _ in range(2):
        ref.zero_grad()
        sh.feature.grad = sh.bias.grad = None
        l1 = ref.loss(batch); l1.backward()
        l2 = sh.loss(batch); l2.backward()
        assert_close(l2.item(), l1.item(), rtol=1e-6, what="loss")
        if kind:
            opt.step()
    if kind:
        assert_close(sh.feature.weight.cpu().numpy(), ref.feature_embedding.weight.detach().cpu().numpy(), what="rows")
        assert_close(sh.bias.weight.cpu().numpy(), ref.bias_embedding.weight.detach().cpu().numpy(), what="bias")
    else:
        assert_close(sh.feature.grad.cpu().numpy(), ref.feature_embedding.weight.grad.cpu().numpy(), what="grad rows")
        assert_close(sh.bias.grad.cpu().numpy(), ref.bias_embedding.weight.grad.cpu().numpy(), what="grad bias")


@pytest.mark.parametrize("kind", ["sgd", "adagrad"])
def test_sharded_fm_world1_clipped_step_matches_fused_optimizer(kind):
    """Global-norm clip on the sharded path (owner-side compact gradients, norm summed over ranks,
    scaled row update) against FusedOptimizer(clip_norm) on the unsharded model."""
    from recsys_amd.models.fields import FieldFM
    from recsys_amd.optim import FusedOptimizer
    from recsys_amd.sharded import ShardedFieldFM
    dims = [7, 1000, 13, 5000, 3, 211]
    torch.manual_seed(2)
    ref = FieldFM(dims, 16).cuda()
    state = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    sh = ShardedFieldFM(dims, 16, "cuda", full_state=state)
    g = torch.Generator().manual_seed(4)
    batch = {"sparse": torch.stack([torch.randint(0, d, (333,), generator=g) for d in dims], 1).cuda(),
             "label": (torch.rand(333, 1, generator=g) < 0.3).double().cuda()}
    sh.calibrate([batch])
    opt = FusedOptimizer(ref, kind=kind, lr=0.1, clip_norm=0.05)
    sh.set_rule(kind, 0.1, clip_norm=0.05)
    for _ in range(2):
        opt.zero_grad()
        l1 = ref.loss(batch); l1.backward(); opt.step()
        l2 = sh.loss(batch); l2.backward(); scale = sh.finish_step()
        assert float(scale) < 1.0
        assert_close(l2.item(), l1.item(), rtol=1e-6, what="loss")
        assert_close(sh.last_total_norm.item(), opt.last_total_norm.item(), rtol=1e-5, what="global norm")
    assert_close(sh.feature.weight.cpu().numpy(), ref.feature_embedding.weight.detach().cpu().numpy(), what="rows")
    assert_close(sh.bias.weight.cpu().numpy(), ref.bias_embedding.weight.detach().cpu().numpy(), what="bias")


def test_sharded_prepare_on_side_stream_is_bitwise_neutral():
    """The local index work of batch i+1 issued on a side stream before step i is launched gives the
    same weights, bit for bit, as doing it inside the step."""
    from recsys_amd.sharded import ShardedFieldFM
    dims = [7, 1000, 13, 5000, 3, 211]
    g = torch.Generator().manual_seed(9)
    batches = [{"sparse": torch.stack([torch.randint(0, d, (500,), generator=g) for d in dims], 1).cuda(),
                "label": (torch.rand(500, 1, generator=g) < 0.3).double().cuda()} for _ in range(3)]
    torch.manual_seed(5)
    state = {"feature_embedding.weight": torch.randn(sum(dims), 16), "bias_embedding.weight": torch.randn(sum(dims), 1)}
    out = []
    for piped in (False, True):
        sh = ShardedFieldFM(dims, 16, "cuda", full_state=state)
        sh.calibrate(batches)
        sh.set_rule("sgd", 0.1, clip_norm=0.05)
        nxt = sh.prepare(batches[0]) if piped else batches[0]
        for i in range(6):
            cur = nxt
            nxt = sh.prepare(batches[(i + 1) % 3]) if piped else batches[(i + 1) % 3]
            assert ("_rx_shard_prep" in cur) == piped
            loss = sh.loss(cur); loss.backward(); sh.finish_step()
        out.append((sh.feature.weight.clone(), sh.bias.weight.clone(), float(loss.item())))
    assert out[0][2] == out[1][2]
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


def test_sharded_prepare_static_overwrites_in_place_and_steps_like_unprepared_batches():
    """prepare_static(batch, out=earlier): every tensor of the earlier result keeps its address and
    holds what a fresh preparation of the new batch holds; steps on batches prepared that way leave
    the same weights, bit for bit, as steps on unprepared batches; a batch of another shape is refused."""
    from recsys_amd.sharded import ShardedFieldFM
    dims = [7, 1000, 13, 5000, 3, 211]
    g = torch.Generator().manual_seed(11)
    batches = [{"sparse": torch.stack([torch.randint(0, d, (400,), generator=g) for d in dims], 1).cuda(),
                "label": (torch.rand(400, 1, generator=g) < 0.3).double().cuda()} for _ in range(3)]
    torch.manual_seed(5)
    state = {"feature_embedding.weight": torch.randn(sum(dims), 16), "bias_embedding.weight": torch.randn(sum(dims), 1)}

    def tensors(prep):
        si, packed, done, flat, slot_of, occ_slot = prep
        assert done is None
        nu = int(si.n_uniq.item())        # per-distinct-row outputs are defined for the first n_uniq entries only
        whole = [si.sorted_pos, si.sorted_slot, si.n_uniq, si.inverse, packed[0], flat, occ_slot]
        return whole + [si.seg_off[:nu + 1], packed[1][:nu], slot_of[:nu], si.uniq_rows[:nu]], si

    out = []
    for static in (False, True):
        sh = ShardedFieldFM(dims, 16, "cuda", full_state=state)
        sh.calibrate(batches)
        sh.set_rule("sgd", 0.1, clip_norm=0.05)
        slot = sh.prepare_static(batches[0]) if static else None
        if static:
            where = [t.data_ptr() for t in tensors(slot["_rx_shard_prep"])[0]]
            again = sh.prepare_static(batches[1], out=slot)
            held, _ = tensors(slot["_rx_shard_prep"])
            assert again is slot and [t.data_ptr() for t in held] == where
            fresh, _ = tensors(sh.prepare_static(batches[1])["_rx_shard_prep"])
            assert all(torch.equal(a, b) for a, b in zip(held, fresh))
            short = {k: v[:100] for k, v in batches[2].items()}
            with pytest.raises(ValueError):
                sh.prepare_static(short, out=slot)
        for i in range(6):
            b = batches[i % 3]
            if static:
                sh.prepare_static(b, out=slot)
                b = dict(b, _rx_shard_prep=slot["_rx_shard_prep"])
            loss = sh.loss(b); loss.backward(); sh.finish_step()
        out.append((sh.feature.weight.clone(), sh.bias.weight.clone(), float(loss.item())))
    assert out[0][2] == out[1][2]
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


@pytest.mark.parametrize("fixed", [False, True])
@pytest.mark.parametrize("shape", ["6_fields_d16", "26_fields_d128"])
def test_sharded_deepfm_world1_matches_field_deepfm(fixed, shape):
    """Sharded tables + replicated BatchNorm tower at world size 1 against the unsharded DeepFM,
    both through the HIP kernels: losses, table rows and tower weights after two SGD steps.
    ``26_fields_d128`` is the model of the 1 B-row configuration (26 tables, 128 columns, 13 dense
    features: a 3344-3344-1672-836-1 tower by the reference's width rule) on small tables."""
    from recsys_amd import synthetic
    from recsys_amd.models.fields import FieldDeepFM
    from recsys_amd.optim import FusedOptimizer
    from recsys_amd.sharded import ShardedFieldDeepFM
    if shape == "6_fields_d16":
        dims, D, n_dense = [7, 1000, 13, 5000, 3, 211], 16, 5
    else:
        dims, D, n_dense = synthetic.criteo_field_dims(total_rows=200_000), 128, 13
    torch.manual_seed(2)
    ref = FieldDeepFM(dims, D, n_dense=n_dense, dropout=0.0).cuda()
    if D == 128:
        assert ref.deep_mlp[0].in_features == 3344 and ref.deep_mlp[4].in_features == 3344 and ref.deep_mlp[8].in_features == 1672
        with torch.no_grad():      # N(0, 1) rows of width 128 over 26 fields saturate every logit: scale them down
            ref.feature_embedding.weight.mul_(0.05)
    state = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    sh = ShardedFieldDeepFM(dims, D, "cuda", n_dense=n_dense, dropout=0.0, full_state=state)
    ref.train(); sh.train()
    g = torch.Generator().manual_seed(4)
    batch = {"sparse": torch.stack([torch.randint(0, d, (333,), generator=g) for d in dims], 1).cuda(),
             "dense": torch.randn(333, n_dense, generator=g).cuda(),
             "label": (torch.rand(333, 1, generator=g) < 0.3).double().cuda()}
    if fixed:
        sh.calibrate([batch])
    clip = 0.05 if fixed else None      # the fixed-capacity case also runs the clipped two-phase step
    opt = FusedOptimizer(ref, kind="sgd", lr=0.1, clip_norm=clip)
    sh.set_rule("sgd", 0.1, clip_norm=clip)
    for _ in range(2):
        opt.zero_grad()
        sh.zero_dense_grads()
        l1 = ref.loss(batch); l1.backward(); opt.step()
        l2 = sh.loss(batch); l2.backward(); sh.allreduce_dense_grads()
        sh.dense_sgd_step(0.1, sh.finish_step(sh.dense_grads()))
        assert_close(l2.item(), l1.item(), rtol=1e-6 if D == 16 else 1e-5, what="loss")
    assert_close(sh.feature.weight.cpu().numpy(), ref.feature_embedding.weight.detach().cpu().numpy(), what="rows")
    assert_close(sh.bias.weight.cpu().numpy(), ref.bias_embedding.weight.detach().cpu().numpy(), what="bias")
    for (k, a), (_, b) in zip(sh.deep_mlp.named_parameters(), ref.deep_mlp.named_parameters()):
        assert_close(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-5, what="tower " + k, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["NeuMF", "TemporalMF"])
def test_pipelined_graph_with_presorted_indices_matches_eager_bitwise(name):
    """ML-1M style models: the graph of batch i sorts the user / item (/ occupation / timestamp)
    indices of batch i+1 on a side stream and the backward picks the pinned result up; cycling over
    3 resident batches gives the same weights, bit for bit, as eager steps on the same sequence."""
    import torch
    from recsys_amd import ops, synthetic
    from recsys_amd.graph import GraphedTrainStep
    from recsys_amd.models.matrix_factorization import TemporalMF
    from recsys_amd.models.neuralcf import NeuMF
    from recsys_amd.optim import FusedOptimizer
    d = synthetic.ML1M
    results = []
    for graphed in (False, True):
        torch.manual_seed(5)
        with torch.device("cuda"):
            if name == "NeuMF":
                m = NeuMF(d["n_users"], d["n_items"], 16, dropout=0.0)
            else:
                m = TemporalMF(d["n_users"], d["n_items"], d["n_occupations"], d["max_timestamp_rank"], 16)
        m.train()
        batches = [synthetic.ml1m_batch(256, seed=20 + i, device="cuda") for i in range(3)]
        opt = FusedOptimizer(m, kind="sgd", lr=0.05, clip_norm=1.0)
        ops.set_index_check("deferred")
        try:
            if graphed:
                gs = GraphedTrainStep(m, opt, warmup=2, pipeline=True)
                gs.add_batches(batches)
                assert "_rx_sorted" in gs.batches[0], "the model did not prepare its index lists"
                for i in range(7):
                    loss = gs(i % 3)
            else:
                for i in [0, 0] + [i % 3 for i in range(7)]:
                    opt.zero_grad(); loss = m.loss(batches[i]); loss.backward(); opt.step()
            ops.raise_if_index_error()
        finally:
            ops.set_index_check("sync")
        results.append(({k: v.detach().clone() for k, v in m.state_dict().items()}, float(loss.item())))
    (sa, la), (sb, lb) = results
    assert la == lb
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize("name", ["FieldFM", "NeuMF"])
def test_round_graph_matches_eager_bitwise(name):
    """One graph holding a whole round over 3 resident batches (index sorts running ahead on the
    side stream), mixed with single-step replays: same weights, bit for bit, as eager steps."""
    from recsys_amd import ops, synthetic
    from recsys_amd.graph import GraphedTrainStep
    from recsys_amd.models.fields import FieldFM
    from recsys_amd.models.neuralcf import NeuMF
    from recsys_amd.optim import FusedOptimizer
    d = synthetic.ML1M
    dims = [7, 1000, 13, 5000, 3, 211]
    results = []
    for graphed in (False, True):
        torch.manual_seed(6)
        with torch.device("cuda"):
            m = FieldFM(dims, 16) if name == "FieldFM" else NeuMF(d["n_users"], d["n_items"], 16, dropout=0.0)
        m.train()
        if name == "FieldFM":
            batches = [synthetic.criteo_batch(dims, 300, seed=40 + i, device="cuda") for i in range(3)]
        else:
            batches = [synthetic.ml1m_batch(256, seed=20 + i, device="cuda") for i in range(3)]
        opt = FusedOptimizer(m, kind="sgd", lr=0.05, clip_norm=1.0)
        ops.set_index_check("deferred")
        try:
            if graphed:
                gs = GraphedTrainStep(m, opt, warmup=2, pipeline=True, cycle=True)
                gs.add_batches(batches)
                gs(0); gs(1); gs(2)                 # single-step graphs
                gs.run_cycle(); gs.run_cycle()      # two rounds
                gs(0)                               # and back
                loss = gs.losses[0]
            else:
                for i in [0, 0] + [0, 1, 2] * 3 + [0]:
                    opt.zero_grad(); loss = m.loss(batches[i]); loss.backward(); opt.step()
            ops.raise_if_index_error()
        finally:
            ops.set_index_check("sync")
        results.append(({k: v.detach().clone() for k, v in m.state_dict().items()}, float(loss.item())))
    (sa, la), (sb, lb) = results
    assert la == lb
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize("ahead", [1, 2])
@pytest.mark.parametrize("name", ["FieldFM_line_rows", "FieldFM", "VanillaMF"])
def test_segment_graphs_match_eager_bitwise(name, ahead):
    """The launch structure bench.py times: ``cycle_len`` < number of resident batches, i.e. one graph per
    SEGMENT of consecutive slots (here 3 + 3 + 2 over 8 batches) with the index preparation of the next
    slot running ahead on the side stream across segment boundaries, mixed with single-step replays.
    Three rounds in cyclic slot order give the same weights, bit for bit, as eager steps."""
    from recsys_amd import ops, synthetic
    from recsys_amd.graph import GraphedTrainStep
    from recsys_amd.models.fields import FieldFM
    from recsys_amd.models.matrix_factorization import VanillaMF
    from recsys_amd.optim import FusedOptimizer
    d = synthetic.ML1M
    dims = [7, 1000, 13, 5000, 3, 211]
    K = 8
    results = []
    for graphed in (False, True):
        torch.manual_seed(6)
        with torch.device("cuda"):
            if name.startswith("FieldFM"):
                m = FieldFM(dims, 16, line_rows=name.endswith("line_rows"))
                batches = [synthetic.criteo_batch(dims, 300, seed=40 + i, device="cuda") for i in range(K)]
            else:
                m = VanillaMF(d["n_users"], d["n_items"], 16)
                batches = [synthetic.ml1m_batch(256, seed=20 + i, device="cuda") for i in range(K)]
        m.train()
        opt = FusedOptimizer(m, kind="sgd", lr=0.05, clip_norm=1.0)
        ops.set_index_check("deferred")
        try:
            if graphed:
                gs = GraphedTrainStep(m, opt, warmup=2, pipeline=True, cycle=True, cycle_len=3, ahead=ahead)
                gs.add_batches(batches)
                assert sorted(gs.segments) == [0, 3, 6] and [gs.segment_at(s) for s in (0, 3, 6)] == [3, 3, 2]
                assert gs.segment_at(1) == 0
                gs(0); gs(1); gs(2)                                   # first pass over the slots: single steps, then segments
                gs.run_cycle(3); gs.run_cycle(6)
                gs.run_cycle(0); gs(3); gs(4); gs(5); gs.run_cycle(6)  # second pass: mixed
                gs.run_cycle(0); gs.run_cycle(3); gs(6); gs(7)         # third pass
                loss = gs.losses[7]
            else:
                for i in [0, 0] + list(range(K)) * 3:
                    opt.zero_grad(); loss = m.loss(batches[i]); loss.backward(); opt.step()
            ops.raise_if_index_error()
        finally:
            ops.set_index_check("sync")
        results.append(({k: v.detach().clone() for k, v in m.state_dict().items()}, float(loss.item())))
    (sa, la), (sb, lb) = results
    assert la == lb
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize("ahead", [1, 2])
@pytest.mark.parametrize("name", ["FieldFM", "VanillaMF"])
def test_refreshed_slots_are_prepared_ahead_bitwise(name, ahead):
    """A stream of 24 DIFFERENT batches through 8 resident slots: the step on slot i is recorded beside the index
    preparation of slot i + ahead, so a slot's new contents (copy_batch) have to be in place before the step
    ``ahead`` slots earlier is launched -- and then the preparation must be the one of the new contents.  Segments
    of three steps and single-step graphs mixed; weights and last loss equal eager steps over the stream, bit
    for bit (a stale preparation would reduce the gradients of the wrong rows)."""
    from recsys_amd import ops, synthetic
    from recsys_amd.graph import GraphedTrainStep
    from recsys_amd.models.fields import FieldFM
    from recsys_amd.models.matrix_factorization import VanillaMF
    from recsys_amd.optim import FusedOptimizer
    d = synthetic.ML1M
    dims = [7, 1000, 13, 5000, 3, 211]
    K, T = 8, 24
    results = []
    for graphed in (False, True):
        torch.manual_seed(6)
        with torch.device("cuda"):
            if name == "FieldFM":
                m = FieldFM(dims, 16)
                stream = [synthetic.criteo_batch(dims, 300, seed=140 + i, device="cuda") for i in range(T)]
            else:
                m = VanillaMF(d["n_users"], d["n_items"], 16)
                stream = [synthetic.ml1m_batch(256, seed=120 + i, device="cuda") for i in range(T)]
        m.train()
        opt = FusedOptimizer(m, kind="sgd", lr=0.05, clip_norm=1.0)
        ops.set_index_check("deferred")
        try:
            if graphed:
                slots = [{k: v.clone() for k, v in b.items()} for b in stream[:K]]
                init = {k: v.detach().clone() for k, v in m.state_dict().items()}
                gs = GraphedTrainStep(m, opt, warmup=2, pipeline=True, cycle=True, cycle_len=3, ahead=ahead)
                gs.add_batches(slots)
                with torch.no_grad():       # the warm-up steps moved the weights: start both runs from the same ones
                    for k, v in m.state_dict().items():
                        v.copy_(init[k])
                j = 0
                plan = [3, 3, 1, 1] + [3, 1, 1, 1, 2] + [1, 1, 1, 3, 2]      # launches per round: 8 steps each
                for L in plan:
                    for t in range(j + ahead, j + L + ahead):       # what the launches below prepare
                        if K <= t < T:
                            gs.copy_batch(t % K, stream[t])
                    if L == 1:
                        gs(j % K)
                    else:
                        assert gs.segment_at(j % K) == L
                        gs.run_cycle(j % K)
                    j += L
                assert j == T
                loss = gs.losses[(T - 1) % K] if plan[-1] == 1 else gs.segments[6][2][-1]
            else:
                for b in stream:
                    opt.zero_grad(); loss = m.loss(b); loss.backward(); opt.step()
            ops.raise_if_index_error()
        finally:
            ops.set_index_check("sync")
        results.append(({k: v.detach().clone() for k, v in m.state_dict().items()}, float(loss.item())))
    (sa, la), (sb, lb) = results
    assert la == lb
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_recording_with_three_concurrent_branches_is_refused(monkeypatch):
    """graph.py replays recordings of at most two concurrent branches (three-wide ones crashed the host
    inside hipGraphLaunch now and then).  NeuMF already runs its GMF half on a second stream; with the
    next batch's index sorts forced onto a third one the recording must be refused, and the same
    model records fine with those sorts behind the GMF half."""
    from recsys_amd import graph, ops, synthetic
    from recsys_amd.models.neuralcf import NeuMF
    from recsys_amd.optim import FusedOptimizer
    d = synthetic.ML1M
    torch.manual_seed(5)
    with torch.device("cuda"):
        m = NeuMF(d["n_users"], d["n_items"], 16, dropout=0.0)
    m.train()
    batches = [synthetic.ml1m_batch(256, seed=20 + i, device="cuda") for i in range(3)]
    opt = FusedOptimizer(m, kind="sgd", lr=0.05, clip_norm=1.0)
    ops.set_index_check("deferred")
    try:
        gs = graph.GraphedTrainStep(m, opt, warmup=1, pipeline=True)
        gs.add_batches(batches)
        shape = graph.recorded_shape(gs.graphs[0])
        assert shape["widest_level"] == 2 and shape["roots"] == 1 and shape["nodes"] > 20
        monkeypatch.setenv("RX_GRAPH_OWN_FORK", "1")
        with pytest.raises(RuntimeError, match="concurrent branches"):
            graph.GraphedTrainStep(m, opt, warmup=0, pipeline=True).add_batches(batches)
        torch.cuda.synchronize()
    finally:
        ops.set_index_check("sync")


def test_recording_refuses_a_live_autograd_graph_of_an_earlier_step():
    """An eager step's loss kept in a variable keeps the parameters' gradient-accumulation nodes alive on the
    default stream; a recording made then would run them there from inside the capture, which crashes the
    host inside hipStreamEndCapture on this runtime (profiles/stale_autograd_probe.py).  The recorder must
    say so instead, and record fine once the old graph is gone."""
    from recsys_amd import graph, ops, synthetic
    from recsys_amd.models.matrix_factorization import VanillaMF
    from recsys_amd.optim import FusedOptimizer
    d = synthetic.ML1M
    torch.manual_seed(5)
    with torch.device("cuda"):
        m = VanillaMF(d["n_users"], d["n_items"], 16)
    m.train()
    batches = [synthetic.ml1m_batch(256, seed=20 + i, device="cuda") for i in range(3)]
    opt = FusedOptimizer(m, kind="dense_adam", lr=1e-2, clip_norm=1.0)
    ops.set_index_check("deferred")
    try:
        opt.zero_grad(); kept = m.loss(batches[0]); kept.backward(); opt.step()
        gs = graph.GraphedTrainStep(m, opt, warmup=0)
        with pytest.raises(RuntimeError, match="autograd graph of an earlier step"):
            gs.add_batch({k: v.clone() for k, v in batches[1].items()})
        del kept
        slot = gs.add_batch({k: v.clone() for k, v in batches[1].items()})
        gs(slot); gs(slot)
        assert gs.losses[slot].grad_fn is None          # what the recorder keeps is detached
        ops.raise_if_index_error()
    finally:
        ops.set_index_check("sync")


def test_sharded_fm_one_rank_through_rccl():
    """A one-rank process group on the nccl (RCCL) backend with the collectives forced on: the
    all-to-all calls, their split-size handling and dtypes run for real on the GPU, and the step
    still matches the unsharded model."""
    import os
    import torch.distributed as dist
    from recsys_amd.models.fields import FieldFM
    from recsys_amd.optim import FusedOptimizer
    from recsys_amd.sharded import ShardedFieldFM
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        dims = [7, 1000, 13, 5000, 3, 211]
        torch.manual_seed(2)
        ref = FieldFM(dims, 16).cuda()
        state = {k: v.detach().clone() for k, v in ref.state_dict().items()}
        sh = ShardedFieldFM(dims, 16, "cuda", full_state=state)
        sh.feature.force_exchange = sh.bias.force_exchange = True
        g = torch.Generator().manual_seed(4)
        batch = {"sparse": torch.stack([torch.randint(0, d, (333,), generator=g) for d in dims], 1).cuda(),
                 "label": (torch.rand(333, 1, generator=g) < 0.3).double().cuda()}
        opt = FusedOptimizer(ref, kind="sgd", lr=0.1)
        sh.set_rule("sgd", 0.1)
        for step in range(3):
            if step == 1:
                sh.calibrate([batch])     # from here on: fixed-capacity exchange, equal all-to-all splits
            ref.zero_grad()
            l1 = ref.loss(batch); l1.backward(); opt.step()
            l2 = sh.loss(batch); l2.backward()
            assert_close(l2.item(), l1.item(), rtol=1e-6, what="loss")
        ops_mod = __import__("recsys_amd.ops", fromlist=["x"])
        ops_mod.raise_if_index_error()
        assert_close(sh.feature.weight.cpu().numpy(), ref.feature_embedding.weight.detach().cpu().numpy(), what="rows")
        assert_close(sh.bias.weight.cpu().numpy(), ref.bias_embedding.weight.detach().cpu().numpy(), what="bias")
        # the DeepFM variant: same exchange plus the tower-gradient all-reduce through RCCL
        from recsys_amd.models.fields import FieldDeepFM
        from recsys_amd.sharded import ShardedFieldDeepFM
        torch.manual_seed(3)
        ref = FieldDeepFM(dims, 16, n_dense=5, dropout=0.0).cuda()
        state = {k: v.detach().clone() for k, v in ref.state_dict().items()}
        sh = ShardedFieldDeepFM(dims, 16, "cuda", n_dense=5, dropout=0.0, full_state=state)
        sh.feature.force_exchange = sh.bias.force_exchange = True
        ref.train(); sh.train()
        batch["dense"] = torch.randn(333, 5, generator=g).cuda()
        sh.calibrate([batch])
        opt = FusedOptimizer(ref, kind="sgd", lr=0.1, clip_norm=0.05)
        sh.set_rule("sgd", 0.1, clip_norm=0.05)
        for step in range(2):
            opt.zero_grad(); sh.zero_dense_grads()
            l1 = ref.loss(batch); l1.backward(); opt.step()
            l2 = sh.loss(batch); l2.backward(); sh.allreduce_dense_grads()
            sh.dense_sgd_step(0.1, sh.finish_step(sh.dense_grads()))
            assert_close(l2.item(), l1.item(), rtol=1e-6, what="deepfm loss")
        ops_mod.raise_if_index_error()
        assert_close(sh.feature.weight.cpu().numpy(), ref.feature_embedding.weight.detach().cpu().numpy(), what="deepfm rows")
        for (k, a), (_, b) in zip(sh.deep_mlp.named_parameters(), ref.deep_mlp.named_parameters()):
            assert_close(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-5, what="tower " + k)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("name", ["VanillaMF", "NeuMF"])
def test_graph_replay_after_copy_batch_matches_eager_on_the_new_batch(name):
    """A recorded step must hold its own sort kernels: after warm-up steps on a batch (which leave
    memoised sorts behind), record with add_batch, refresh the batch in place with copy_batch and
    replay — the weights must equal, bit for bit, eager steps on the same batch sequence."""
    from recsys_amd import ops, synthetic
    from recsys_amd.graph import GraphedTrainStep
    from recsys_amd.models.matrix_factorization import VanillaMF
    from recsys_amd.models.neuralcf import NeuMF
    from recsys_amd.optim import FusedOptimizer
    d = synthetic.ML1M
    results = []
    for graphed in (False, True):
        torch.manual_seed(9)
        with torch.device("cuda"):
            m = VanillaMF(d["n_users"], d["n_items"], 16) if name == "VanillaMF" else NeuMF(d["n_users"], d["n_items"], 16, dropout=0.0)
        m.train()
        first = synthetic.ml1m_batch(256, seed=70, device="cuda")
        later = [synthetic.ml1m_batch(256, seed=71 + i, device="cuda") for i in range(3)]
        opt = FusedOptimizer(m, kind="sgd", lr=0.05, clip_norm=1.0)
        ops.set_index_check("deferred")
        try:
            if graphed:
                gs = GraphedTrainStep(m, opt, warmup=2)
                slot = gs.add_batch({k: v.clone() for k, v in first.items()})   # 2 eager warm-up steps on `first`
                gs(slot)
                for b in later:
                    gs.copy_batch(slot, b)
                    loss = gs(slot)
            else:
                for b in [first, first, first] + later:
                    opt.zero_grad(); loss = m.loss(b); loss.backward(); opt.step()
            ops.raise_if_index_error()
        finally:
            ops.set_index_check("sync")
        results.append(({k: v.detach().clone() for k, v in m.state_dict().items()}, float(loss.item())))
    (sa, la), (sb, lb) = results
    assert la == lb
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize("ctor_clip,loop_clip", [(None, 1.0), (1.0, None)])
def test_train_step_clip_overrides_the_optimizers_own_setting(ctor_clip, loop_clip):
    """recsys/utils.py:101: the clip belongs to the train loop.  A FusedOptimizer built with another
    setting must behave, from the first step on, like one built with the loop's value."""
    from recsys_amd import synthetic
    from recsys_amd.models.matrix_factorization import BiasMF
    from recsys_amd.optim import FusedOptimizer
    from recsys_amd.utils import train_step
    d = synthetic.ML1M
    states = []
    for clip_at_ctor in (ctor_clip, loop_clip):
        torch.manual_seed(11)
        with torch.device("cuda"):
            m = BiasMF(d["n_users"], d["n_items"], 16)
        m.train()
        opt = FusedOptimizer(m, kind="sgd", lr=0.5, clip_norm=clip_at_ctor)
        for i in range(2):
            train_step(m, opt, synthetic.ml1m_batch(128, seed=90 + i, device="cuda"), loop_clip)
        states.append({k: v.detach().clone() for k, v in m.state_dict().items()})
    for k in states[0]:
        assert torch.equal(states[0][k], states[1][k]), k


def test_pending_row_updates_are_applied_when_the_clip_is_dropped_before_step():
    from recsys_amd import synthetic
    from recsys_amd.models.matrix_factorization import VanillaMF
    from recsys_amd.optim import FusedOptimizer
    d = synthetic.ML1M
    torch.manual_seed(12)
    with torch.device("cuda"):
        m = VanillaMF(d["n_users"], d["n_items"], 16)
    ref = {k: v.detach().clone() for k, v in m.state_dict().items()}
    b = synthetic.ml1m_batch(128, seed=5, device="cuda")
    opt = FusedOptimizer(m, kind="sgd", lr=0.5, clip_norm=1e9)     # a clip that never binds: compact + pending path
    opt.zero_grad(); m.loss(b).backward()
    opt.clip_norm = None
    opt.step()
    after_dropped = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.load_state_dict(ref)
    opt2 = FusedOptimizer(m, kind="sgd", lr=0.5, clip_norm=None)
    opt2.zero_grad(); m.loss(b).backward(); opt2.step()
    for k, v in m.state_dict().items():
        assert not torch.equal(v, ref[k]) or "bias" in k
        assert torch.equal(v, after_dropped[k]), k


def test_pack_by_owner_overflow_raises_and_leaves_tables_untouched():
    """A group larger than the exchange capacity: the entries without a slot are marked, nothing is
    read or written through them, and the deferred check reports the overflow."""
    from recsys_amd import ops
    g = torch.Generator().manual_seed(3)
    n, G, dim = 600, 2, 8
    idx = torch.unique(torch.randint(0, 5000, (n,), generator=g)).cuda()
    n = idx.numel()
    counts, send_local, perm = ops.bucketize_by_owner(idx, G, 5000)
    cap = int(counts.max().item()) - 7          # too small for the larger group
    ops.set_index_check("deferred")
    try:
        packed, src_of = ops.pack_by_owner(send_local, perm, counts, cap, pad_rows=16)
        assert int((src_of < 0).sum().item()) >= 7
        assert int(src_of.max().item()) < G * cap
        rows = torch.randn(G * cap, dim, device="cuda")
        got = ops.permute_rows(rows, src_of, None)
        ok = src_of >= 0
        assert torch.equal(got[ok], rows[src_of[ok].long()])
        assert torch.count_nonzero(got[~ok]) == 0
        sent = torch.full((G * cap, dim), 7.0, device="cuda")
        ops.unpermute_rows(torch.randn(n, dim, device="cuda"), src_of, None, out=sent)
        untouched = torch.ones(G * cap, dtype=torch.bool, device="cuda")
        untouched[src_of[ok].long()] = False
        assert torch.all(sent[untouched] == 7.0)
        with pytest.raises(RuntimeError, match="capacity"):
            ops.raise_if_index_error()
    finally:
        ops.set_index_check("sync")


def test_topk_wider_than_the_kernel_limit_follows_the_reference_order():
    from recsys_amd.utils import topk
    g = torch.Generator().manual_seed(8)
    scores = torch.randint(0, 50, (37, 3706), generator=g).float()     # many ties
    codes = torch.stack([torch.randperm(3706, generator=g) for _ in range(37)])
    # the documented tie order: a stable ascending argsort read backwards (numpy's default argsort is
    # not stable on rows this wide, so the host mirror is only compared where scores are distinct)
    order = np.argsort(scores.numpy(), axis=1, kind="stable")[:, ::-1][:, :10]
    dev = topk(scores.cuda(), array=codes.cuda(), k=10).cpu().numpy()
    assert np.array_equal(dev, np.take_along_axis(codes.numpy(), order, axis=1))
    assert np.array_equal(topk(scores.cuda(), k=10).cpu().numpy(), order)
    distinct = torch.stack([torch.randperm(3706, generator=g) for _ in range(5)]).float()
    assert np.array_equal(topk(distinct.cuda(), k=10).cpu().numpy(), topk(distinct.numpy(), k=10))


# ------------------------------------------------------------------ the reference's captures through the device branches
def _golden_loader_data(rec):
    import scipy.sparse as scs
    shape = tuple(int(x) for x in rec["shape"])
    r, c = rec["implicit_rows"], rec["implicit_cols"]
    return {
        "implicit_matrix": scs.csr_matrix((np.ones(len(r)), (r, c)), shape=shape),
        "timestamp_matrix": scs.csr_matrix((rec["timestamp_vals"].astype(np.float64), (r, c)), shape=shape),
        "test_codes": rec["test_codes"], "test_true": rec["test_codes"][:, :1],
        "test_timestamp_rank": rec["test_timestamp_rank"], "negative_samples": rec["negative_samples"],
        "user_infos": rec["user_infos"],
    }


def test_golden_ranking_and_metrics_through_the_device_branches():
    """The captured scores / codes / predictions / NDCG / HR of the reference's eval path
    (recsys/utils.py:40-75, recsys/metrics.py:4-38) fed as GPU tensors: rx_topk_codes and the
    torch-on-device metric code must give the captured values (ranks exactly, metrics to 1e-12)."""
    from conftest import load_golden
    from recsys_amd.metrics import hr_score, ndcg_score
    from recsys_amd.utils import topk
    rec = load_golden("eval")
    scores = torch.from_numpy(rec["scores"]).cuda()
    codes = torch.from_numpy(rec["codes"]).cuda()
    k = int(rec["k"])
    pred = topk(scores, array=codes, k=k)
    assert pred.is_cuda and np.array_equal(pred.cpu().numpy(), rec["pred"])
    assert np.array_equal(topk(scores, array=rec["codes"], k=k).cpu().numpy(), rec["pred"])   # host codes, device scores
    assert np.array_equal(topk(scores, subset=torch.from_numpy(rec["subset"]).cuda(), k=5).cpu().numpy(), rec["pred_subset_k5"])
    assert np.array_equal(topk(scores, k=3).cpu().numpy(), rec["pred_plain_k3"])
    true = torch.from_numpy(rec["true"]).cuda()
    assert abs(ndcg_score(true, pred) - float(rec["ndcg"])) < 1e-12
    assert abs(hr_score(true, pred) - float(rec["hr"])) < 1e-12
    assert abs(ndcg_score(rec["true"], pred) - float(rec["ndcg"])) < 1e-12                     # host truth, device predictions
    assert abs(hr_score(rec["true"], pred) - float(rec["hr"])) < 1e-12


def test_golden_batches_from_the_device_resident_loader():
    """ImplicitData on the GPU with the reference's numpy sampler under the captured seed: the one
    train batch (as a set of rows: the epoch permutation is torch's) and the first test batch hold
    exactly the captured tensors, dtypes included (recsys/dataset.py:165-187,228-282)."""
    from conftest import group, load_golden
    from recsys_amd.dataset import ImplicitData
    rec = load_golden("loader")
    n = len(rec["implicit_rows"])
    idata = ImplicitData(_golden_loader_data(rec), device="cuda", train_batch_size=n, test_batch_size=4, sampler="numpy")
    np.random.seed(int(rec["seed"]))
    (batch,) = list(idata.create_negative_sampled_train_dataloader())
    assert all(v.is_cuda for v in batch.values())
    host = {k: v.cpu().numpy() for k, v in batch.items()}
    order = np.lexsort((host["item_code"][:, 0], host["user_code"]))
    exp = group(rec, "train_batch")
    assert set(host) == set(exp)
    for k, v in exp.items():
        assert str(batch[k].dtype) == str(rec[f"train_dtype/{k}"]), k
        assert np.array_equal(host[k][order], v), k
    tdl = idata.create_test_dataloader()
    assert len(tdl) == int(rec["n_test_batches"])
    first = next(iter(tdl))
    exp = group(rec, "test_batch0")
    assert set(first) == set(exp) and "label" not in first
    for k, v in exp.items():
        assert first[k].is_cuda and str(first[k].dtype) == str(rec[f"test_dtype/{k}"]), k
        assert np.array_equal(first[k].cpu().numpy(), v), k


def test_golden_sampler_rows_feed_a_model_step_on_the_device():
    """The captured output of the reference sampler (seeded numpy, recsys/dataset.py:165-187) as a
    device batch through a drop-in model: every row is [u, positive, 4 negatives from u's pool] and
    the loss over it equals the CPU oracle's."""
    from conftest import load_golden
    from oracle.models import OracleModel, init_state
    from recsys_amd.models.matrix_factorization import VanillaMF
    rec = load_golden("sampler")
    data, labels = rec["train_data"], rec["train_labels"]
    n_users, n_items = (int(x) for x in rec["shape"])
    pools = rec["negative_samples"]
    for row in data:
        assert set(row[2:].tolist()) <= set(pools[row[0]].tolist())
    batch = {"user_code": torch.from_numpy(data[:, 0].copy()), "item_code": torch.from_numpy(data[:, 1:].copy()),
             "label": torch.from_numpy(labels)}
    state = init_state("VanillaMF", n_users=n_users, n_items=n_items, embedding_dim=16, seed=2)
    oracle = OracleModel("VanillaMF", state, n_users=n_users, n_items=n_items)
    m = VanillaMF(n_users, n_items, 16)
    m.load_state_dict(state)
    m = m.cuda()
    loss = m.loss({k: v.cuda() for k, v in batch.items()})
    assert_close(loss.item(), oracle.loss(batch).item(), rtol=1e-6, what="loss on the sampler capture")


def test_negative_sampler_kernel_stream_and_distribution():
    """rx_sample_negatives: bit for bit the documented counter-based stream (numpy mirror), every
    draw from the user's own pool, positives and labels laid out like recsys/dataset.py:176-185, and
    the pool positions uniform (chi-square over 200 k draws into 50 positions)."""
    from recsys_amd import ops
    g = torch.Generator().manual_seed(8)
    n_users, P, n, n_neg = 300, 50, 40_000, 5
    pool = torch.stack([torch.randperm(5000, generator=g)[:P] for _ in range(n_users)]).contiguous()
    users = torch.randint(0, n_users, (n,), generator=g)
    items = torch.randint(0, 5000, (n,), generator=g)
    out, labels = ops.sample_negatives(users.cuda(), items.cuda(), pool.cuda(), n_neg, seed=1234)
    exp = ops.sample_negatives_host(users.numpy(), items.numpy(), pool.numpy(), n_neg, 1234)
    assert out.dtype == torch.int64 and labels.dtype == torch.float64
    assert np.array_equal(out.cpu().numpy(), exp)
    assert torch.equal(out[:, 0].cpu(), items)
    assert torch.equal(labels[:, 0], torch.ones(n, dtype=torch.float64, device="cuda")) and not labels[:, 1:].any()
    # membership and the position histogram, recovered through each pool's inverse map
    inv = torch.full((n_users, 5000), -1, dtype=torch.int64)
    inv.scatter_(1, pool, torch.arange(P).expand(n_users, P))
    pos = inv[users[:, None].expand(n, n_neg), out[:, 1:].cpu()]
    assert (pos >= 0).all(), "a draw from outside the user's pool"
    counts = torch.bincount(pos.reshape(-1), minlength=P).double()
    expect = n * n_neg / P
    chi2 = float(((counts - expect) ** 2 / expect).sum())
    assert chi2 < 100.0, chi2                    # 49 degrees of freedom: P(chi2 > 100) < 3e-5
    # another seed is another stream; the same seed the same one
    other, _ = ops.sample_negatives(users.cuda(), items.cuda(), pool.cuda(), n_neg, seed=1235, want_labels=False)
    same, _ = ops.sample_negatives(users.cuda(), items.cuda(), pool.cuda(), n_neg, seed=1234, want_labels=False)
    assert torch.equal(same, out) and (other[:, 1:] != out[:, 1:]).float().mean() > 0.9
    # a user id outside the pool table is reported, not read
    bad = users.clone(); bad[7] = n_users
    ops.set_index_check("deferred")
    try:
        ops.sample_negatives(bad.cuda(), items.cuda(), pool.cuda(), n_neg, seed=1)
        with pytest.raises(IndexError):
            ops.raise_if_index_error()
    finally:
        ops.set_index_check("sync")


def test_device_loader_epochs_are_reproducible_from_the_seed():
    from recsys_amd.dataset import ImplicitData, synthetic_ml1m_data
    d = synthetic_ml1m_data(seed=3, n_users=200, n_items=300, n_occupations=7, n_train=6000, pool=40, n_test_codes=20)
    a = ImplicitData(d, device="cuda", train_batch_size=512, seed=5)
    b = ImplicitData(d, device="cuda", train_batch_size=512, seed=5)
    c = ImplicitData(d, device="cpu", train_batch_size=512, seed=5)
    for _ in range(2):
        ea, eb, ec = (x.create_negative_sampled_train_dataloader().tensors for x in (a, b, c))
        assert torch.equal(ea["item_code"], eb["item_code"])
        assert torch.equal(ea["item_code"].cpu(), ec["item_code"])          # the numpy mirror on a CPU device
    e2 = a.create_negative_sampled_train_dataloader().tensors
    assert not torch.equal(e2["item_code"], ea["item_code"])               # next epoch, next stream


@pytest.mark.parametrize("feed", ["loader", "batches"])
@pytest.mark.parametrize("name", ["VanillaMF", "TemporalMF", "NeuMF", "VanillaMF_without_prepare", "VanillaMF_adam", "NeuMF_adam"])
def test_train_epoch_graph_replay_matches_eager_bitwise(name, feed):
    """utils.LoopGraph: two epochs with the step replayed as a graph give the same weights and epoch
    losses, bit for bit, as the eager loop on the same seeded loader -- both when the recorded step
    itself hands over to the next batch of the resident epoch (feed "loader": rx_epoch_advance; the
    epoch ends in a short batch, which runs eagerly) and when every batch is copied in from the host
    (feed "batches": any iterable of batches)."""
    from recsys_amd import ops
    from recsys_amd.dataset import ImplicitData, synthetic_ml1m_data
    from recsys_amd.models import matrix_factorization as mf
    from recsys_amd.optim import FusedOptimizer
    from recsys_amd.utils import LoopGraph, train_epoch
    d = synthetic_ml1m_data(seed=3, n_users=400, n_items=300, n_occupations=7, n_train=12000, pool=60, n_test_codes=50)
    opt_kind, name = ("dense_adam", name[:-5]) if name.endswith("_adam") else ("adagrad", name)   # _adam: the reference's Adam over dense gradients
    results = []
    for graphed in (False, True):
        data = ImplicitData(d, device="cuda", train_batch_size=500, seed=9)
        torch.manual_seed(0)
        kw = dict(n_users=data.n_users, n_items=data.n_items, embedding_dim=16)
        if name == "TemporalMF":
            kw.update(n_occupations=data.n_occupations, max_timestamp_rank=data.max_timestamp_rank)
        if name == "NeuMF":      # two-stream model (GMF half on a branch); no dropout: its masks differ between graph and eager
            from recsys_amd.models.neuralcf import NeuMF
            model = NeuMF(data.n_users, data.n_items, 16, dropout=0.0).cuda().train()
        elif name == "VanillaMF_without_prepare":
            class Unprepared(mf.VanillaMF):
                def _index_lists(self, inputs):
                    return []
            model = Unprepared(**kw).cuda().train()
        else:
            model = getattr(mf, name)(**kw).cuda().train()
        opt = FusedOptimizer(model, kind=opt_kind, lr=0.05 if opt_kind == "adagrad" else 1e-3, clip_norm=1.0)
        step = LoopGraph(model, opt) if graphed else None
        ops.set_index_check("deferred")
        try:
            sums = []
            for _ in range(2):
                dl = data.create_negative_sampled_train_dataloader()
                sums.append(float(train_epoch(model, opt, dl if feed == "loader" else list(dl), 1.0, step).item()))
            ops.raise_if_index_error()
        finally:
            ops.set_index_check("sync")
        if graphed:
            assert (step.epoch_slot if feed == "loader" else step.slot) is not None
            paired = feed == "loader"                         # two static batches take turns
            assert len(step.step.graphs) == (2 if paired else 1) and step.paired == paired
            assert not paired or step.step._pair_multi[1] == 8                  # ... and eight steps replay per launch
            if feed == "loader":
                assert int(step.cursor.item()) == len(dl) - (1 if dl.length % dl.batch_size else 0)
                assert int(step.ticket.item()) == 0
        results.append((sums, {k: v.detach().clone() for k, v in model.state_dict().items()}))
    assert results[0][0] == results[1][0]
    for k, v in results[0][1].items():
        assert torch.equal(v, results[1][1][k]), k


@pytest.mark.parametrize("model", ["fm", "deepfm"])
def test_sharded_step_replayed_as_a_graph_through_rccl_matches_eager_bitwise(model):
    """graph.ShardedGraphStep at one rank on the nccl (RCCL) backend with the exchanges forced on:
    index work, the all-to-alls, backward, clip and update replay as one HIP graph per batch and
    leave the same weights, bit for bit, as eager steps.  Runs in a process of its own (it owns a
    process group and must tear it down after releasing the graphs)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ROWS="300000", DIM="16", B="4096", MODEL=model, MASTER_PORT="29547")
    out = subprocess.run([sys.executable, "-u", os.path.join(root, "profiles", "sharded_graph_probe.py")], env=env,
                         capture_output=True, text=True, timeout=150)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "bitwise equal True" in out.stdout and "group destroyed" in out.stdout, out.stdout[-2000:]
    assert "index work one batch ahead: True" in out.stdout      # graph i prepares batch i+1 into static buffers


@pytest.mark.parametrize("graphed", [False, True])
def test_neumf_second_stream_branch_matches_one_stream_bitwise(graphed, monkeypatch):
    """NeuMF runs its GMF half on a second stream (autograd.Branch), forward and backward; the
    per-occurrence gradients it leaves for the clipped optimizer step and the dense gradients of
    linear_gmf are joined before they are read.  Weights after three steps must equal the
    one-stream run bit for bit, eagerly and replayed as a graph."""
    from recsys_amd import autograd as rxa, ops, synthetic
    from recsys_amd.graph import GraphedTrainStep
    from recsys_amd.models.neuralcf import NeuMF
    from recsys_amd.optim import FusedOptimizer
    batches = [synthetic.ml1m_batch(512, seed=s, device="cuda") for s in (1, 2, 3)]
    out = []
    for off in (True, False):
        monkeypatch.setattr(rxa, "_BRANCHES_OFF", off)
        torch.manual_seed(0)
        m = NeuMF(synthetic.ML1M["n_users"], synthetic.ML1M["n_items"], 32, dropout=0.0).cuda().train()
        opt = FusedOptimizer(m, kind="adagrad", lr=0.05, clip_norm=0.5)
        ops.set_index_check("deferred")
        try:
            if graphed:
                gs = GraphedTrainStep(m, opt, warmup=1)
                for b in batches:
                    gs.add_batch(b)
                for i in range(len(batches)):
                    gs(i)
            else:
                for b in batches:
                    opt.zero_grad()
                    rxa.backward_from_loss(m.loss(b))
                    opt.step()
            torch.cuda.synchronize()
            ops.raise_if_index_error()
        finally:
            ops.set_index_check("sync")
        out.append({k: v.detach().clone() for k, v in m.state_dict().items()})
    for k, v in out[0].items():
        assert torch.equal(v, out[1][k]), k
