"""Golden-vector generator (run in the build container only).

Imports the reference implementation from /root/reference (read-only, never
shipped), drives every model on the embedding/interaction hot path with small
seeded inputs on the CPU, and stores inputs + expected outputs as plain
``.npz`` fixtures (no pickles) next to this script.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference \
        python tests/golden/make_golden.py

What is captured per model (keys are ``<group>/<name>``):

* ``state/*``       initial ``state_dict`` (parameters and BatchNorm buffers)
* ``batch/*``       one train batch with the layout of recsys/dataset.py:255-262
* ``eval_logits``   ``model.eval(); model(batch)``
* ``train_logits``, ``loss``, ``grad/*``  train mode, dropout probability 0
* ``clip_total_norm`` value returned by ``clip_grad_norm_(params, 1.0)``
* ``sgd{1,3}/*``, ``adagrad{1,3}/*``  state after 1 / 3 optimizer steps on the
  same batch (no clipping)
* ``adam{1,3}/*``   state after 1 / 3 steps of the reference train step
  (recsys/utils.py:96-103: zero_grad, loss, backward, clip 1.0, Adam 1e-3)
* ``pw_logits``     GMF / MLP / NeuMF point-wise forward (item_code of shape [B])

Also captured: the negative sampler (recsys/dataset.py:165-187) under a numpy
seed, ``topk`` (recsys/utils.py:40-58) and the metrics (recsys/metrics.py).
"""
import copy
import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import scipy.sparse as scs
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

from recsys.models.matrix_factorization import (  # noqa: E402
    BiasMF,
    FactorizationMachine,
    SideFeaturesMF,
    TemporalMF,
    VanillaMF,
)
from recsys.models.neuralcf import (  # noqa: E402
    GMF,
    MLP,
    DeepFactorizationMachine,
    NeuMF,
    WidenDeep,
)
from recsys import dataset as ref_dataset  # noqa: E402
from recsys import metrics as ref_metrics  # noqa: E402
from recsys import utils as ref_utils  # noqa: E402

N_USERS, N_ITEMS, N_OCC, MAX_TS = 50, 40, 5, 30
B, N = 16, 5

SIDE = dict(n_occupations=N_OCC, max_timestamp_rank=MAX_TS)
MODELS = {
    "VanillaMF": (VanillaMF, {}),
    "BiasMF": (BiasMF, {}),
    "SideFeaturesMF": (SideFeaturesMF, {"n_occupations": N_OCC}),
    "TemporalMF": (TemporalMF, SIDE),
    "FactorizationMachine": (FactorizationMachine, SIDE),
    "GMF": (GMF, {}),
    "MLP": (MLP, {}),
    "NeuMF": (NeuMF, {}),
    "WidenDeep": (WidenDeep, SIDE),
    "DeepFactorizationMachine": (DeepFactorizationMachine, SIDE),
}
POINTWISE = {"GMF", "MLP", "NeuMF"}


def make_batch(seed, b=B, n=N):
    g = torch.Generator().manual_seed(seed)
    users = torch.randint(0, N_USERS, (b,), generator=g)
    # repeat some users / items inside the batch so dedup has work to do
    users[1] = users[0]
    items = torch.randint(0, N_ITEMS, (b, n), generator=g)
    items[2, 0] = items[3, 1]
    items[4] = items[5]
    occ = torch.randint(0, N_OCC, (b,), generator=g)
    ts = torch.randint(0, MAX_TS, (b,), generator=g)
    label = torch.zeros(b, n, dtype=torch.float64)
    label[:, 0] = 1
    return {
        "user_code": users,
        "item_code": items,
        "user_occupation": occ,
        "item_timestamp_rank": ts,
        "label": label,
    }


def zero_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0


def sd_np(model):
    return {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}


def run_steps(model0, batch, make_opt, steps, clip):
    model = copy.deepcopy(model0)
    model.train()
    opt = make_opt(model.parameters())
    out = {}
    for s in range(1, max(steps) + 1):
        opt.zero_grad()
        loss = model.loss(batch)
        loss.backward()
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
        opt.step()
        if s in steps:
            out[s] = sd_np(model)
    return out


def capture_model(name, cls, extra, dim, seed):
    torch.manual_seed(seed)
    model = cls(n_users=N_USERS, n_items=N_ITEMS, embedding_dim=dim, **extra)
    zero_dropout(model)
    batch = make_batch(seed + 1000)
    rec = {}
    for k, v in sd_np(model).items():
        rec[f"state/{k}"] = v
    for k, v in batch.items():
        rec[f"batch/{k}"] = v.numpy()

    model.eval()
    with torch.no_grad():
        rec["eval_logits"] = model(batch).numpy()
        if name in POINTWISE:
            pw = dict(batch)
            pw["item_code"] = batch["item_code"][:, 0].contiguous()
            rec["pw_logits"] = model(pw, pointwise=True).numpy()

    m = copy.deepcopy(model)
    m.train()
    logits = m(batch)
    rec["train_logits"] = logits.detach().numpy()
    m2 = copy.deepcopy(model)
    m2.train()
    loss = m2.loss(batch)
    assert loss.dtype == torch.float64
    rec["loss"] = loss.detach().numpy()
    loss.backward()
    for k, p in m2.named_parameters():
        assert not p.grad.is_sparse
        rec[f"grad/{k}"] = p.grad.detach().clone().numpy()
    total = torch.nn.utils.clip_grad_norm_(m2.parameters(), 1.0)
    rec["clip_total_norm"] = total.detach().numpy()

    steps = (1, 3)
    for tag, mk, clip in (
        ("sgd", lambda p: torch.optim.SGD(p, lr=0.1), None),
        ("adagrad", lambda p: torch.optim.Adagrad(p, lr=0.1), None),
        ("adam", lambda p: torch.optim.Adam(p, lr=1e-3), 1.0),
    ):
        res = run_steps(model, batch, mk, steps, clip)
        for s, sd in res.items():
            for k, v in sd.items():
                rec[f"{tag}{s}/{k}"] = v
    return rec


def capture_sampler():
    """recsys/dataset.py:165-187 under a fixed numpy seed on a toy matrix."""
    rng = np.random.RandomState(3)
    n_u, n_i, pool = 12, 30, 7
    dense = (rng.rand(n_u, n_i) < 0.2).astype(np.float64)
    dense[:, 0] = 0
    dense[0, 1] = 1  # every row pattern is fine; keep at least one positive
    mat = scs.csr_matrix(dense)
    negative_samples = rng.randint(0, n_i, size=(n_u, pool))
    np.random.seed(1234)
    train_data, train_labels = ref_dataset.create_negative_sampled_train_data(
        mat, negative_samples, 4
    )
    rows, cols = mat.nonzero()
    return {
        "implicit_rows": rows.astype(np.int64),
        "implicit_cols": cols.astype(np.int64),
        "shape": np.array([n_u, n_i], dtype=np.int64),
        "negative_samples": negative_samples.astype(np.int64),
        "seed": np.array(1234, dtype=np.int64),
        "n_negatives": np.array(4, dtype=np.int64),
        "train_data": train_data.astype(np.int64),
        "train_labels": train_labels,
    }


def capture_loader():
    """Batch dict layout and dtypes out of ImplicitData + DataLoader
    (recsys/dataset.py:204-291) on toy data, shuffle made inert by batch == length."""
    rng = np.random.RandomState(5)
    n_u, n_i, pool = 9, 20, 6
    dense = (rng.rand(n_u, n_i) < 0.25).astype(np.float64)
    dense[np.arange(n_u), rng.randint(0, n_i, n_u)] = 1
    implicit = scs.csr_matrix(dense)
    ts_dense = np.zeros_like(dense)
    r, c = dense.nonzero()
    ts_dense[r, c] = rng.randint(1, 15, size=r.shape[0])
    ts = scs.csr_matrix(ts_dense)
    negative_samples = rng.randint(0, n_i, size=(n_u, pool))
    test_codes = rng.randint(0, n_i, size=(n_u, 10))
    user_infos = np.stack(
        [np.arange(n_u), rng.randint(0, 2, n_u), rng.randint(0, 7, n_u), rng.randint(0, 4, n_u)], 1
    )
    data = {
        "implicit_matrix": implicit,
        "timestamp_matrix": ts,
        "test_codes": test_codes,
        "test_true": test_codes[:, :1],
        "test_timestamp_rank": rng.randint(0, 15, n_u),
        "negative_samples": negative_samples,
        "user_infos": user_infos,
    }
    length = int(implicit.nnz)
    idata = ref_dataset.ImplicitData(data, device="cpu", train_batch_size=length, test_batch_size=4)
    np.random.seed(77)
    dl = idata.create_negative_sampled_train_dataloader()
    (batch,) = list(dl)
    out = {
        "implicit_rows": r.astype(np.int64),
        "implicit_cols": c.astype(np.int64),
        "timestamp_vals": ts_dense[r, c].astype(np.int64),
        "shape": np.array([n_u, n_i], dtype=np.int64),
        "negative_samples": negative_samples.astype(np.int64),
        "test_codes": test_codes.astype(np.int64),
        "test_timestamp_rank": data["test_timestamp_rank"].astype(np.int64),
        "user_infos": user_infos.astype(np.int64),
        "seed": np.array(77, dtype=np.int64),
        "n_occupations": np.array(idata.n_occupations, dtype=np.int64),
        "max_timestamp_rank": np.array(idata.max_timestamp_rank, dtype=np.int64),
    }
    # the loader shuffles; store the batch sorted by (user, positive item) so the
    # comparison is order independent
    order = np.lexsort((batch["item_code"][:, 0].numpy(), batch["user_code"].numpy()))
    for k, v in batch.items():
        out[f"train_batch/{k}"] = v.numpy()[order]
        out[f"train_dtype/{k}"] = np.array(str(v.dtype))
    tdl = idata.create_test_dataloader()
    tb = list(tdl)
    out["n_test_batches"] = np.array(len(tb), dtype=np.int64)
    for k, v in tb[0].items():
        out[f"test_batch0/{k}"] = v.numpy()
        out[f"test_dtype/{k}"] = np.array(str(v.dtype))
    return out


def capture_eval():
    """topk (recsys/utils.py:40-58) + NDCG / HR (recsys/metrics.py:4-38)."""
    rng = np.random.RandomState(11)
    n_u, n_c, k = 37, 100, 10
    scores = rng.randn(n_u, n_c).astype(np.float32)
    codes = np.stack([rng.permutation(500)[:n_c] for _ in range(n_u)]).astype(np.int64)
    true = codes[:, :1].copy()
    pred = ref_utils.topk(scores, array=codes, k=k)
    with redirect_stdout(io.StringIO()):
        ndcg, hr = ref_metrics.compute_metrics(true, pred)
    sub = np.stack([rng.permutation(n_c)[:20] for _ in range(n_u)]).astype(np.int64)
    pred_sub = ref_utils.topk(scores, subset=sub, k=5)
    pred_plain = ref_utils.topk(scores, k=3)
    return {
        "scores": scores,
        "codes": codes,
        "true": true,
        "k": np.array(k, dtype=np.int64),
        "pred": pred.astype(np.int64),
        "ndcg": np.array(ndcg, dtype=np.float64),
        "hr": np.array(hr, dtype=np.float64),
        "subset": sub,
        "pred_subset_k5": pred_sub.astype(np.int64),
        "pred_plain_k3": pred_plain.astype(np.int64),
    }


def main():
    torch.set_num_threads(1)
    torch.use_deterministic_algorithms(True)
    total = 0
    for i, (name, (cls, extra)) in enumerate(MODELS.items()):
        for dim in (16, 64) if name in ("VanillaMF", "NeuMF", "FactorizationMachine") else (16,):
            rec = capture_model(name, cls, extra, dim, seed=100 + i)
            rec["dims"] = np.array([N_USERS, N_ITEMS, N_OCC, MAX_TS, dim, B, N], dtype=np.int64)
            fn = os.path.join(HERE, f"{name}_d{dim}.npz")
            np.savez_compressed(fn, **rec)
            total += os.path.getsize(fn)
            print(f"{fn}: {len(rec)} arrays, {os.path.getsize(fn)} bytes")
    for tag, fnc in (("sampler", capture_sampler), ("loader", capture_loader), ("eval", capture_eval)):
        rec = fnc()
        fn = os.path.join(HERE, f"{tag}.npz")
        np.savez_compressed(fn, **rec)
        total += os.path.getsize(fn)
        print(f"{fn}: {len(rec)} arrays, {os.path.getsize(fn)} bytes")
    print("total bytes", total)
    with open(os.path.join(HERE, "PROVENANCE.txt"), "w") as f:
        f.write(
            "Generated by tests/golden/make_golden.py from the reference imported at /root/reference\n"
            f"torch {torch.__version__} (CPU), numpy {np.__version__}, python {sys.version.split()[0]}\n"
        )


if __name__ == "__main__":
    main()
