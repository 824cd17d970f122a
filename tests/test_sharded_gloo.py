"""Row-sharded tables + all-to-all exchange, world_size 2 over gloo on the CPU.  The kernels are
replaced by the CPU stand-ins of oracle/cpu_ops.py (injected as ``backend``); what is under test
is the exchange logic of recsys_amd/sharded.py: owner bucketing, split sizes, permutations,
gradient routing and owner-side updates, against a single-process run on the whole batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

DIMS = [5, 40, 3, 17, 9]
DIM = 8
B_LOCAL = 24


def _full_state(seed=0):
    g = torch.Generator().manual_seed(seed)
    n = sum(DIMS)
    return {"feature_embedding.weight": torch.randn(n, DIM, generator=g),
            "bias_embedding.weight": torch.randn(n, 1, generator=g)}


def _global_batch(world, seed=1):
    g = torch.Generator().manual_seed(seed)
    B = B_LOCAL * world
    sparse = torch.stack([torch.randint(0, d, (B,), generator=g) for d in DIMS], 1)
    return {"sparse": sparse, "label": (torch.rand(B, 1, generator=g) < 0.3).double()}


def _worker(rank, world, port, kind, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import cpu_ops
        from recsys_amd.sharded import ShardedFieldFM, ShardedTable
        torch.manual_seed(0)
        full = _full_state()
        batch = _global_batch(world)
        sl = slice(rank * B_LOCAL, (rank + 1) * B_LOCAL)
        local = {k: v[sl] for k, v in batch.items()}
        # plain lookup equals indexing the full table
        t = ShardedTable(sum(DIMS), DIM, "cpu", backend=cpu_ops, full_weight=full["feature_embedding.weight"])
        idx = torch.randint(0, sum(DIMS), (37 + rank,), generator=torch.Generator().manual_seed(5 + rank))
        got = t.gather(t.plan(idx))
        assert torch.equal(got, full["feature_embedding.weight"][idx])
        # two data-parallel steps
        m = ShardedFieldFM(DIMS, DIM, "cpu", backend=cpu_ops, full_state=full)
        fixed = "+capacity" in kind
        clip = 0.05 if "+clip" in kind else None     # small enough to bite on every step
        kind = kind.split("+")[0]
        if fixed:   # fixed-capacity exchange: equal all-to-all splits, capacity from the batch itself
            assert m.calibrate([local]) % 256 == 0
        if kind != "dense":
            m.set_rule(kind, lr=0.1, clip_norm=clip)
        losses = []
        for _ in range(2):
            m.feature.grad = m.bias.grad = None
            loss = m.loss(local)
            (loss / world).backward()
            scale = m.finish_step()
            assert (scale is None) == (clip is None) and (clip is None or float(scale) < 1.0)
            losses.append(loss.detach())
            if kind == "dense":
                m.feature.weight -= 0.1 * m.feature.grad
                m.bias.weight -= 0.1 * m.bias.grad
        gathered = [None] * world
        dist.all_gather_object(gathered, (m.feature.weight, m.bias.weight, torch.stack(losses)))
        if rank == 0:
            torch.save(gathered, out)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("kind", ["sgd", "adagrad", "dense", "sgd+capacity", "adagrad+capacity", "sgd+clip",
                                  "adagrad+capacity+clip"])
def test_two_rank_sharded_fm_matches_single_process(tmp_path, kind):
    from oracle.fields import FieldOracle
    world = 2
    out = str(tmp_path / "shards.pt")
    mp.spawn(_worker, args=(world, _free_port(), kind, out), nprocs=world, join=True)
    shards = torch.load(out, weights_only=False)
    n = sum(DIMS)
    feat, bias = torch.empty(n, DIM), torch.empty(n, 1)
    for r, (fw, bw, _) in enumerate(shards):
        feat[r::world], bias[r::world] = fw, bw
    clip = 0.05 if "+clip" in kind else None
    kind = kind.split("+")[0]
    oracle = FieldOracle("FieldFM", _full_state(), DIMS)
    opt = (torch.optim.Adagrad if kind == "adagrad" else torch.optim.SGD)(oracle.parameters(), lr=0.1)
    batch = _global_batch(world)
    ref_losses = []
    for _ in range(2):
        opt.zero_grad()
        loss = oracle.loss(batch)
        loss.backward()
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(oracle.parameters(), clip)
        opt.step()
        ref_losses.append(loss.item())
    mean_losses = torch.stack([s[2] for s in shards]).mean(0)
    assert torch.allclose(mean_losses, torch.tensor(ref_losses, dtype=mean_losses.dtype), rtol=1e-6)
    sd = oracle.state_dict()
    assert torch.allclose(feat, sd["feature_embedding.weight"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(bias, sd["bias_embedding.weight"], rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ DeepFM: sharded tables + data-parallel tower
N_DENSE = 5


def _deepfm_state(seed=3):
    from recsys_amd.models.fields import FieldDeepFM
    torch.manual_seed(seed)
    m = FieldDeepFM(DIMS, DIM, n_dense=N_DENSE, dropout=0.0)
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def _deep_batch(world, seed=2):
    b = _global_batch(world, seed)
    g = torch.Generator().manual_seed(seed + 100)
    b["dense"] = torch.randn(B_LOCAL * world, N_DENSE, generator=g)
    return b


def _deep_worker(rank, world, port, fixed, out, clip=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import cpu_ops
        from recsys_amd.sharded import ShardedFieldDeepFM
        full = _deepfm_state()
        batch = _deep_batch(world)
        sl = slice(rank * B_LOCAL, (rank + 1) * B_LOCAL)
        local = {k: v[sl] for k, v in batch.items()}
        m = ShardedFieldDeepFM(DIMS, DIM, "cpu", n_dense=N_DENSE, dropout=0.0, backend=cpu_ops, full_state=full)
        m.train()
        if fixed:
            m.calibrate([local])
        m.set_rule("sgd", lr=0.1, clip_norm=clip)
        losses = []
        for _ in range(2):
            m.zero_dense_grads()
            loss = m.loss(local)
            (loss / world).backward()
            m.allreduce_dense_grads()
            m.dense_sgd_step(0.1, m.finish_step(m.dense_grads()))
            losses.append(loss.detach())
        dense = {k: v.detach().clone() for k, v in m.deep_mlp.state_dict().items()}
        gathered = [None] * world
        dist.all_gather_object(gathered, (m.feature.weight, m.bias.weight, torch.stack(losses), dense))
        if rank == 0:
            torch.save(gathered, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fixed,clip", [(False, None), (True, None), (True, 0.05)])
def test_two_rank_sharded_deepfm_matches_single_process(tmp_path, fixed, clip):
    """Tables sharded, tower replicated: two steps on two ranks equal two steps of the CPU
    restatement on the mean of the per-rank losses (every rank's BatchNorm sees its own slice)."""
    from oracle.fields import FieldOracle
    world = 2
    out = str(tmp_path / "deep_shards.pt")
    mp.spawn(_deep_worker, args=(world, _free_port(), fixed, out, clip), nprocs=world, join=True)
    shards = torch.load(out, weights_only=False)
    n = sum(DIMS)
    feat, bias = torch.empty(n, DIM), torch.empty(n, 1)
    for r, (fw, bw, _, _) in enumerate(shards):
        feat[r::world], bias[r::world] = fw, bw
    oracle = FieldOracle("FieldDeepFM", _deepfm_state(), DIMS, dropout=0.0)
    opt = torch.optim.SGD(oracle.parameters(), lr=0.1)
    batch = _deep_batch(world)
    halves = [{k: v[r * B_LOCAL:(r + 1) * B_LOCAL] for k, v in batch.items()} for r in range(world)]
    ref_losses = []
    for _ in range(2):
        opt.zero_grad()
        per_rank = [oracle.loss(h) for h in halves]
        (sum(per_rank) / world).backward()
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(oracle.parameters(), clip)
        opt.step()
        ref_losses.append([l.item() for l in per_rank])
    got = torch.stack([s[2] for s in shards]).T      # [step, rank]
    assert torch.allclose(got, torch.tensor(ref_losses, dtype=got.dtype), rtol=1e-6)
    sd = oracle.state_dict()
    assert torch.allclose(feat, sd["feature_embedding.weight"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(bias, sd["bias_embedding.weight"], rtol=1e-5, atol=1e-6)
    for r in range(world):
        for k, v in shards[r][3].items():
            if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                continue
            assert torch.allclose(v, sd["deep_mlp." + k], rtol=1e-5, atol=1e-6), (r, k)
