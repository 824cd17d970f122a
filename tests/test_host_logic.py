"""Host-side mirrors of the reference's batching, ranking and metrics against the golden captures
(CPU only)."""
import numpy as np
import scipy.sparse as scs
import torch

from conftest import group, load_golden


def test_negative_sampler_matches_reference_under_numpy_seed():
    from recsys_amd.dataset import create_negative_sampled_train_data
    rec = load_golden("sampler")
    shape = tuple(int(x) for x in rec["shape"])
    mat = scs.csr_matrix((np.ones(len(rec["implicit_rows"])), (rec["implicit_rows"], rec["implicit_cols"])), shape=shape)
    np.random.seed(int(rec["seed"]))
    data, labels = create_negative_sampled_train_data(mat, rec["negative_samples"], int(rec["n_negatives"]))
    assert np.array_equal(data, rec["train_data"])
    assert labels.dtype == np.float64 and np.array_equal(labels, rec["train_labels"])


def _loader_data(rec):
    shape = tuple(int(x) for x in rec["shape"])
    r, c = rec["implicit_rows"], rec["implicit_cols"]
    return {
        "implicit_matrix": scs.csr_matrix((np.ones(len(r)), (r, c)), shape=shape),
        "timestamp_matrix": scs.csr_matrix((rec["timestamp_vals"].astype(np.float64), (r, c)), shape=shape),
        "test_codes": rec["test_codes"], "test_true": rec["test_codes"][:, :1],
        "test_timestamp_rank": rec["test_timestamp_rank"], "negative_samples": rec["negative_samples"],
        "user_infos": rec["user_infos"],
    }


def test_batch_dict_layout_and_dtypes_match_reference_loader():
    from recsys_amd.dataset import ImplicitData
    rec = load_golden("loader")
    data = _loader_data(rec)
    n = len(rec["implicit_rows"])
    idata = ImplicitData(data, device="cpu", train_batch_size=n, test_batch_size=4, sampler="numpy")
    assert idata.n_occupations == int(rec["n_occupations"])
    assert idata.max_timestamp_rank == int(rec["max_timestamp_rank"])
    np.random.seed(int(rec["seed"]))
    dl = idata.create_negative_sampled_train_dataloader()
    assert len(dl) == 1
    (batch,) = list(dl)
    order = np.lexsort((batch["item_code"][:, 0].numpy(), batch["user_code"].numpy()))
    exp = group(rec, "train_batch")
    assert set(batch) == set(exp)
    for k, v in exp.items():
        assert str(batch[k].dtype) == str(rec[f"train_dtype/{k}"]), k
        assert np.array_equal(batch[k].numpy()[order], v), k
    tdl = idata.create_test_dataloader()
    assert idata.test_dataloader is tdl and len(tdl) == int(rec["n_test_batches"])
    first = next(iter(tdl))
    exp = group(rec, "test_batch0")
    assert set(first) == set(exp) and "label" not in first
    for k, v in exp.items():
        assert str(first[k].dtype) == str(rec[f"test_dtype/{k}"]), k
        assert np.array_equal(first[k].numpy(), v), k


def test_device_sampler_draws_from_each_users_pool():
    from recsys_amd.dataset import ImplicitData
    rec = load_golden("loader")
    idata = ImplicitData(_loader_data(rec), device="cpu", train_batch_size=5, sampler="device")
    dl = idata.create_negative_sampled_train_dataloader(n_negatives=3)
    seen = 0
    for b in dl:
        assert b["item_code"].shape[1] == 4 and b["label"].dtype == torch.float64
        assert torch.equal(b["label"][:, 0], torch.ones(len(b["label"]), dtype=torch.float64))
        assert not b["label"][:, 1:].any()
        for u, items in zip(b["user_code"].tolist(), b["item_code"].tolist()):
            assert set(items[1:]) <= set(rec["negative_samples"][u].tolist())
        seen += len(b["user_code"])
    assert seen == len(rec["implicit_rows"]) and len(dl) == -(-seen // 5)


def test_topk_and_metrics_match_reference():
    from recsys_amd.metrics import compute_metrics, hr_score, ndcg_score
    from recsys_amd.utils import topk
    rec = load_golden("eval")
    pred = topk(rec["scores"], array=rec["codes"], k=int(rec["k"]))
    assert np.array_equal(pred, rec["pred"])
    assert np.array_equal(topk(rec["scores"], subset=rec["subset"], k=5), rec["pred_subset_k5"])
    assert np.array_equal(topk(rec["scores"], k=3), rec["pred_plain_k3"])
    assert abs(ndcg_score(rec["true"], rec["pred"]) - float(rec["ndcg"])) < 1e-12
    assert abs(hr_score(rec["true"], rec["pred"]) - float(rec["hr"])) < 1e-12
    # several held-out items per user
    true2 = np.concatenate([rec["true"], rec["codes"][:, 5:6]], 1)
    rel = np.array([np.isin(rec["pred"][i], true2[i]) for i in range(len(true2))])
    disc = np.log2(np.arange(2, rec["pred"].shape[1] + 2))
    irel = np.zeros(rec["pred"].shape[1]); irel[:2] = 1
    exp = ((rel / disc).sum(1) / (irel / disc).sum()).mean()
    assert abs(ndcg_score(true2, rec["pred"]) - exp) < 1e-12
    assert abs(hr_score(true2, rec["pred"]) - np.array([np.isin(true2[i], rec["pred"][i]) for i in range(len(true2))]).mean()) < 1e-12


def test_synthetic_ml1m_shape():
    from recsys_amd.dataset import synthetic_ml1m_data
    d = synthetic_ml1m_data(seed=1, n_users=200, n_items=300, n_train=8000, pool=50, n_test_codes=20)
    assert d["implicit_matrix"].shape == (200, 300)
    assert d["negative_samples"].shape == (200, 50) and d["test_codes"].shape == (200, 20)
    seen = d["implicit_matrix"].toarray() > 0
    for u in range(200):
        assert not seen[u, d["negative_samples"][u]].any()      # pool = unseen items
        assert not seen[u, d["test_true"][u, 0]]                 # held-out positive is not in train
    assert (np.asarray(d["implicit_matrix"].sum(1)).reshape(-1) >= 19).all()


def test_leave_one_out_split_matches_reference_on_toy_files(tmp_path):
    """recsys/dataset.py:10-156 on a toy ratings.dat / users.dat pair (users shuffled in the file, tied
    timestamps, movie ids whose string order differs from the numeric one, fewer movies than the
    500-item pool): every array of the captured split, values and dtypes."""
    from recsys_amd.dataset import ImplicitData, load_ml1m_data, split_data_loo
    rec = load_golden("split")
    rp, up, out = tmp_path / "ratings.dat", tmp_path / "users.dat", tmp_path / "ml1m.npz"
    rp.write_text("\n".join(rec["ratings_lines"].tolist()) + "\n")
    up.write_text("\n".join(rec["users_lines"].tolist()) + "\n")
    data = split_data_loo(rp, up, n_test_codes=int(rec["n_test_codes"]), out=out)
    for k in ("rating_matrix", "implicit_matrix", "timestamp_matrix"):
        m = data[k].tocoo()
        order = np.lexsort((m.col, m.row))
        exp = group(rec, k)
        assert tuple(m.shape) == tuple(exp["shape"]) and str(data[k].dtype) == str(exp["dtype"]), k
        assert np.array_equal(m.row[order], exp["rows"]) and np.array_equal(m.col[order], exp["cols"]), k
        assert np.array_equal(m.data[order], exp["vals"]), k
    for k in ("test_codes", "test_true", "test_timestamp_rank", "negative_samples", "user_infos"):
        assert np.array_equal(np.asarray(data[k]), rec[k]), k
        assert str(np.asarray(data[k]).dtype) == str(rec[f"{k}_dtype"]), k
    with np.testing.assert_raises(ValueError):
        load_ml1m_data(out)                                   # a pickled dict: only on request
    again = load_ml1m_data(out, trust_pickle=True)
    assert np.array_equal(again["test_codes"], data["test_codes"])
    idata = ImplicitData(data, device="cpu", train_batch_size=16, sampler="numpy")
    assert idata.n_users == 14 and len(idata.create_negative_sampled_train_dataloader()) == -(-data["implicit_matrix"].nnz // 16)
