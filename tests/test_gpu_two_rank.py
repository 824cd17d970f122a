"""Two to four ranks with the real HIP kernels.  A one-GPU machine cannot host two RCCL ranks, so both
processes open the same card and the collectives travel over gloo; everything else — owner
buckets, fixed-capacity slots, owner-side sort / reduce / update, the global clip, the tower
all-reduce — is the code that runs under RCCL on a multi-GPU node.  Checked against the CPU
restatement stepping on the whole batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

DIMS = [7, 1000, 13, 5000, 3, 211, 40000]
DIM = 16
B_LOCAL = 500
N_DENSE = 5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fm_state(seed=0):
    g = torch.Generator().manual_seed(seed)
    n = sum(DIMS)
    return {"feature_embedding.weight": torch.randn(n, DIM, generator=g),
            "bias_embedding.weight": torch.randn(n, 1, generator=g)}


def _deepfm_state(seed=3):
    from recsys_amd.models.fields import FieldDeepFM
    torch.manual_seed(seed)
    m = FieldDeepFM(DIMS, DIM, n_dense=N_DENSE, dropout=0.0)
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def _global_batch(world, seed=1):
    g = torch.Generator().manual_seed(seed)
    B = B_LOCAL * world
    # skewed draws: hot rows are shared by both ranks, so the owners really sum over sources
    sparse = torch.stack([(torch.rand(B, generator=g) ** 3 * d).long().clamp_(max=d - 1) for d in DIMS], 1)
    return {"sparse": sparse, "label": (torch.rand(B, 1, generator=g) < 0.3).double(),
            "dense": torch.randn(B, N_DENSE, generator=g)}


def _stage_collectives_through_host():
    """gloo's own device all-to-all stalls now and then between processes that share a card (every
    rank inside the same call, the device idle), so the buffers cross gloo as host tensors."""
    native = dist.all_to_all_single

    def staged(out, inp, output_split_sizes=None, input_split_sizes=None, group=None):
        if not out.is_cuda:
            return native(out, inp, output_split_sizes, input_split_sizes, group=group)
        o = torch.empty(out.shape, dtype=out.dtype)
        native(o, inp.cpu(), output_split_sizes, input_split_sizes, group=group)
        out.copy_(o)
    dist.all_to_all_single = staged


def _worker(rank, world, port, model, kind, out):
    import faulthandler
    faulthandler.dump_traceback_later(120, exit=True)     # a stuck collective names itself
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        _stage_collectives_through_host()
        from recsys_amd import ops
        from recsys_amd.sharded import ShardedFieldDeepFM, ShardedFieldFM, ShardedTable
        batch = _global_batch(world)
        sl = slice(rank * B_LOCAL, (rank + 1) * B_LOCAL)
        local = {k: v[sl].cuda() for k, v in batch.items()}
        fixed = "+capacity" in kind
        clip = 0.05 if "+clip" in kind else None
        rule = kind.split("+")[0]
        if model == "fm":
            full = _fm_state()
            t = ShardedTable(sum(DIMS), DIM, "cuda", full_weight=full["feature_embedding.weight"])
            idx = torch.randint(0, sum(DIMS), (3001 + 17 * rank,), generator=torch.Generator().manual_seed(5 + rank))
            got = t.gather(t.plan(idx.cuda()))
            assert torch.equal(got.cpu(), full["feature_embedding.weight"][idx]), "sharded lookup differs from indexing"
            m = ShardedFieldFM(DIMS, DIM, "cuda", full_state=full)
        else:
            m = ShardedFieldDeepFM(DIMS, DIM, "cuda", n_dense=N_DENSE, dropout=0.0, full_state=_deepfm_state())
            m.train()
        if fixed:
            m.calibrate([local])
        m.set_rule(rule, lr=0.1, clip_norm=clip)
        losses = []
        for _ in range(2):
            local = m.prepare(local) if fixed and "_rx_shard_prep" not in local else {k: v for k, v in local.items() if k != "_rx_shard_prep"}
            if model == "fm":
                loss = m.loss(local)
                (loss / world).backward()
                m.finish_step()
            else:
                m.zero_dense_grads()
                loss = m.loss(local)
                (loss / world).backward()
                m.allreduce_dense_grads()
                m.dense_sgd_step(0.1, m.finish_step(m.dense_grads()))
            losses.append(loss.detach().cpu())
        ops.raise_if_index_error()
        dense = {} if model == "fm" else {k: v.detach().cpu() for k, v in m.deep_mlp.state_dict().items()}
        gathered = [None] * world
        dist.all_gather_object(gathered, (m.feature.weight.cpu(), m.bias.weight.cpu(), torch.stack(losses), dense))
        if rank == 0:
            torch.save(gathered, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("model,kind,world", [("fm", "sgd", 2), ("fm", "sgd+capacity+clip", 2), ("fm", "adagrad+capacity", 2),
                                              ("fm", "rowwise_adagrad+capacity", 2), ("deepfm", "sgd+capacity+clip", 2),
                                              ("fm", "sgd+capacity+clip", 4), ("deepfm", "sgd+capacity", 3)])
def test_ranks_on_one_card_match_whole_batch_step(tmp_path, model, kind, world):
    from oracle.fields import FieldOracle
    out = str(tmp_path / "shards.pt")
    mp.spawn(_worker, args=(world, _free_port(), model, kind, out), nprocs=world, join=True)
    shards = torch.load(out, weights_only=False)
    n = sum(DIMS)
    feat, bias = torch.empty(n, DIM), torch.empty(n, 1)
    for r, s in enumerate(shards):
        feat[r::world], bias[r::world] = s[0], s[1]
    clip = 0.05 if "+clip" in kind else None
    rule = kind.split("+")[0]
    if model == "fm":
        oracle = FieldOracle("FieldFM", _fm_state(), DIMS)
    else:
        oracle = FieldOracle("FieldDeepFM", _deepfm_state(), DIMS, dropout=0.0)
    opt = (torch.optim.Adagrad if rule == "adagrad" else torch.optim.SGD)(oracle.parameters(), lr=0.1)
    batch = _global_batch(world)
    halves = [{k: v[r * B_LOCAL:(r + 1) * B_LOCAL] for k, v in batch.items()} for r in range(world)]
    offs = torch.tensor([0] + DIMS[:-1]).cumsum(0)
    touched = torch.zeros(n, dtype=torch.bool)
    touched[(batch["sparse"] + offs[None, :]).reshape(-1)] = True
    row_state = {k: torch.zeros(n) for k in ("feature_embedding.weight", "bias_embedding.weight")}
    ref_losses = []
    for _ in range(2):
        opt.zero_grad()
        per_rank = [oracle.loss(h) for h in halves]      # every rank's BatchNorm sees its own slice
        (sum(per_rank) / world).backward()
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(oracle.parameters(), clip)
        if rule == "rowwise_adagrad":
            # no torch counterpart: state[r] += mean_d g[r, d]^2, w[r] -= lr g[r] / (sqrt(state[r]) + eps), touched rows only
            with torch.no_grad():
                for k, st in row_state.items():
                    p = oracle.params[k]
                    st += (p.grad ** 2).mean(1)
                    upd = 0.1 * p.grad / (st.sqrt()[:, None] + 1e-10)
                    p -= torch.where(touched[:, None], upd, torch.zeros_like(upd))
        else:
            opt.step()
        ref_losses.append([l.item() for l in per_rank])
    got = torch.stack([s[2] for s in shards]).T          # [step, rank]
    assert torch.allclose(got, torch.tensor(ref_losses, dtype=got.dtype), rtol=1e-5)
    sd = oracle.state_dict()
    assert torch.allclose(feat, sd["feature_embedding.weight"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(bias, sd["bias_embedding.weight"], rtol=1e-5, atol=1e-6)
    for r in range(world):
        for k, v in shards[r][3].items():
            if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                continue
            assert torch.allclose(v, sd["deep_mlp." + k], rtol=2e-5, atol=1e-6), (r, k)
