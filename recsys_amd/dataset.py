"""Batching for the implicit-feedback pipelines, device resident.

Mirrors the batch contract of recsys/dataset.py (``ImplicitData`` with
``create_negative_sampled_train_dataloader(n_negatives=4)`` / ``create_test_dataloader()`` /
``test_dataloader``; batches are dicts with ``user_code [B] i64``, ``item_code [B, 1+n_neg] i64``,
``user_occupation [B] i64``, ``item_timestamp_rank [B] i64``, ``label [B, 1+n_neg] f64``; test
batches carry no label, recsys/dataset.py:255-262,276-282) without its per-sample Python
``__getitem__`` + collate: the positives and their side features are uploaded once, each epoch's
negatives are drawn, the epoch is permuted once, and batches are contiguous slices of tensors
that already sit in HBM.

``sampler="numpy"`` draws the negatives with the reference's exact numpy calls
(recsys/dataset.py:165-187), so a numpy seed reproduces its ``train_data`` bit for bit;
``sampler="device"`` draws them on the GPU with one launch of rx_sample_negatives (same
distribution, a counter-based stream of its own: epoch e of a loader seeded s is reproducible from
(s, e)); on a CPU device the same rule runs in numpy.
"""
from __future__ import annotations

import math
from typing import Dict, Iterator, Optional

import numpy as np
import scipy.sparse as scs
import torch


def create_negative_sampled_train_data(train_inputs, negative_samples, n_negatives):
    """Host sampler with the reference's call sequence: for every positive (u, i), ``n_negatives``
    draws ``negative_samples[u, randint(0, pool)]``; rows are [u, i_pos, neg_1..neg_n], labels
    [1, 0, ..., 0] in float64."""
    rows, cols = train_inputs.nonzero()
    neg_rows = rows.repeat(n_negatives)
    neg_cols = np.random.randint(0, negative_samples.shape[1], neg_rows.shape[0])
    negatives = negative_samples[neg_rows, neg_cols].reshape(-1, n_negatives)
    data = np.hstack([np.vstack([rows, cols]).T, negatives])
    labels = np.zeros((data.shape[0], n_negatives + 1), dtype=float)
    labels[:, 0] = 1
    return data, labels


class EpochLoader:
    """Iterable over dict batches that are slices of epoch tensors resident on ``device``.
    ``len()`` is the number of batches (last one may be short, like DataLoader's default)."""

    def __init__(self, tensors: Dict[str, torch.Tensor], batch_size: int, shuffle: bool,
                 generator: Optional[torch.Generator] = None):
        self.tensors, self.batch_size, self.shuffle, self.generator = tensors, batch_size, shuffle, generator
        self.length = next(iter(tensors.values())).shape[0]

    def __len__(self):
        return math.ceil(self.length / self.batch_size)

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        tensors = self.tensors
        if self.shuffle:
            dev = next(iter(tensors.values())).device
            perm = torch.randperm(self.length, device=dev, generator=self.generator)
            tensors = {k: v.index_select(0, perm) for k, v in tensors.items()}
        for lo in range(0, self.length, self.batch_size):
            yield {k: v[lo:lo + self.batch_size] for k, v in tensors.items()}


class ImplicitData:
    """Same constructor contract as recsys/dataset.py:204-226: ``data`` is the dict written by the
    reference's split (implicit_matrix, timestamp_matrix, test_codes, test_true,
    test_timestamp_rank, negative_samples, user_infos)."""

    def __init__(self, data: dict, device="cpu", train_batch_size: int = 512, test_batch_size: int = 1024,
                 sampler: str = "device", seed: int = 0):
        if sampler not in ("device", "numpy"):
            raise ValueError(sampler)
        self.implicit_matrix = data["implicit_matrix"]
        self.timestamp_matrix = data["timestamp_matrix"]
        self.test_codes = data["test_codes"]
        self.test_true = data["test_true"]
        self.test_timestamp_rank = data["test_timestamp_rank"]
        self.negative_samples = data["negative_samples"]
        self.user_infos = data["user_infos"]
        self.n_users, self.n_items = self.implicit_matrix.shape
        self.n_occupations = np.unique(self.user_infos[:, -1]).shape[0]
        self.max_timestamp_rank = int(self.timestamp_matrix.data.max() + 2)
        self.device = torch.device(device)
        self.train_batch_size, self.test_batch_size, self.sampler = train_batch_size, test_batch_size, sampler
        self._test_dataloader = None
        # positives and their side features never change between epochs: upload them once
        rows, cols = self.implicit_matrix.nonzero()
        ts = np.asarray(self.timestamp_matrix[rows, cols]).reshape(-1).astype(np.int64)
        dev = self.device
        self._rows = torch.from_numpy(rows.astype(np.int64)).to(dev)
        self._cols = torch.from_numpy(cols.astype(np.int64)).to(dev)
        self._occ = torch.from_numpy(self.user_infos[rows, -1].astype(np.int64)).to(dev)
        self._ts = torch.from_numpy(ts).to(dev)
        self._pool = torch.from_numpy(np.ascontiguousarray(self.negative_samples).astype(np.int64)).to(dev)
        self._gen = torch.Generator(device=dev)
        self._gen.manual_seed(seed)
        self.seed, self._epoch = seed, 0

    def create_negative_sampled_train_dataloader(self, n_negatives: int = 4) -> EpochLoader:
        n = self._rows.shape[0]
        if self.sampler == "numpy":
            data, labels = create_negative_sampled_train_data(self.implicit_matrix, self.negative_samples, n_negatives)
            items = torch.from_numpy(np.ascontiguousarray(data[:, 1:]).astype(np.int64)).to(self.device)
            label = torch.from_numpy(labels).to(self.device)
        else:
            epoch_seed = (self.seed << 20) + self._epoch        # one stream per (loader seed, epoch)
            self._epoch += 1
            if self.device.type == "cuda":
                from . import ops
                items, label = ops.sample_negatives(self._rows, self._cols, self._pool, n_negatives, epoch_seed)
            else:
                from .ops import sample_negatives_host
                items = torch.from_numpy(sample_negatives_host(self._rows.numpy(), self._cols.numpy(), self._pool.numpy(),
                                                               n_negatives, epoch_seed))
                label = torch.zeros(n, n_negatives + 1, dtype=torch.float64)
                label[:, 0] = 1
        tensors = {"user_code": self._rows, "item_code": items, "user_occupation": self._occ,
                   "item_timestamp_rank": self._ts, "label": label}
        return EpochLoader(tensors, self.train_batch_size, shuffle=True, generator=self._gen)

    def create_test_dataloader(self) -> EpochLoader:
        if self._test_dataloader is None:
            dev = self.device
            users = np.arange(self.n_users)
            tensors = {
                "user_code": torch.from_numpy(users.astype(np.int64)).to(dev),
                "item_code": torch.from_numpy(np.ascontiguousarray(self.test_codes).astype(np.int64)).to(dev),
                "user_occupation": torch.from_numpy(self.user_infos[users, -1].astype(np.int64)).to(dev),
                "item_timestamp_rank": torch.from_numpy(np.asarray(self.test_timestamp_rank).astype(np.int64)).to(dev),
            }
            self._test_dataloader = EpochLoader(tensors, self.test_batch_size, shuffle=False)
        return self._test_dataloader

    @property
    def test_dataloader(self):
        return self._test_dataloader


def synthetic_ml1m_data(seed: int = 0, n_users: int = 6040, n_items: int = 3706, n_occupations: int = 21,
                        n_train: int = 994_169, pool: int = 500, n_test_codes: int = 100) -> dict:
    """A ``data`` dict with MovieLens-1M's shape (the real ratings.dat is not redistributable and is
    absent from the reference checkout): Zipf-like item popularity, at least 20 interactions per
    user, a top-``pool`` popular-unseen negative pool per user, 1 held-out positive + 99 sampled
    negatives as test codes (the split the reference builds in recsys/dataset.py:99-156)."""
    rng = np.random.RandomState(seed)
    per_user = np.maximum(20, rng.zipf(1.6, n_users) * 20)
    per_user = np.minimum(per_user, n_items // 2)
    per_user = np.round(per_user * (n_train + n_users) / per_user.sum()).astype(np.int64)
    per_user = np.clip(per_user, 20, n_items // 2)
    pop = 1.0 / np.arange(1, n_items + 1) ** 0.9
    pop /= pop.sum()
    rows, cols, ts = [], [], []
    test_true = np.zeros((n_users, 1), dtype=np.int64)
    test_ts = np.zeros(n_users, dtype=np.int64)
    seen = []
    for u in range(n_users):
        items = rng.choice(n_items, size=int(per_user[u]), replace=False, p=pop)
        seen.append(items)
        test_true[u, 0] = items[-1]
        test_ts[u] = len(items) - 1
        rows.append(np.full(len(items) - 1, u)); cols.append(items[:-1]); ts.append(np.arange(len(items) - 1))
    rows, cols, ts = np.concatenate(rows), np.concatenate(cols), np.concatenate(ts)
    implicit = scs.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(n_users, n_items))
    timestamps = scs.csr_matrix((ts.astype(np.float64), (rows, cols)), shape=(n_users, n_items))
    popularity = np.asarray(implicit.sum(0)).reshape(-1)
    order = np.argsort(-popularity, kind="stable")
    negative_samples = np.zeros((n_users, pool), dtype=np.int64)
    for u in range(n_users):
        mask = np.ones(n_items, dtype=bool)
        mask[seen[u]] = False
        negative_samples[u] = order[mask[order]][:pool]
    negs = np.take_along_axis(negative_samples, rng.randint(0, pool, (n_users, n_test_codes - 1)), 1)
    user_infos = np.stack([np.arange(n_users), rng.randint(0, 2, n_users), rng.randint(0, 7, n_users),
                           rng.randint(0, n_occupations, n_users)], 1)
    user_infos[:n_occupations, -1] = np.arange(n_occupations)
    return {"implicit_matrix": implicit, "timestamp_matrix": timestamps, "test_codes": np.hstack([test_true, negs]),
            "test_true": test_true, "test_timestamp_rank": test_ts, "negative_samples": negative_samples,
            "user_infos": user_infos}


def _read_dat(path, n_cols):
    with open(path, encoding="ISO-8859-1") as f:
        rows = [line.rstrip("\n").split("::") for line in f if line.strip()]
    if any(len(r) != n_cols for r in rows):
        raise ValueError(f"{path}: expected {n_cols} '::'-separated columns per line")
    return [list(c) for c in zip(*rows)] if rows else [[] for _ in range(n_cols)]


def _string_codes(values):
    """Codes of a pandas ``category`` column read from text: the rank of each value among the
    distinct values **as strings** ("10" sorts before "9"), in the smallest signed integer type."""
    cats, codes = np.unique(np.asarray(values, dtype=str), return_inverse=True)
    dtype = np.int8 if len(cats) <= 127 else np.int16 if len(cats) <= 32767 else np.int32
    return codes.astype(dtype)


def _rank_first_within(group, key, descending=False):
    """pandas ``groupby(group)[key].rank(method="first") - 1``: position of every element inside its
    group ordered by key, ties in order of appearance (for descending too)."""
    n = len(group)
    order = np.lexsort((np.arange(n), -key if descending else key, group))
    g = group[order]
    start = np.flatnonzero(np.r_[True, g[1:] != g[:-1]])
    first = np.repeat(start, np.diff(np.r_[start, n]))
    rank = np.empty(n, dtype=np.float64)
    rank[order] = np.arange(n) - first
    return rank


def split_data_loo(ratings_path, users_path, n_test_codes: int = 100, out=None, pool: int = 500, seed: int = 47) -> dict:
    """The reference's leave-one-out split (recsys/dataset.py:10-156) from ``ratings.dat`` /
    ``users.dat`` ("::"-separated MovieLens files), without pandas: user code = id - 1, movie code =
    rank of the id string among the rated ids, each user's latest rating (ties: first in the file)
    held out, the per-user negative pool = the ``pool`` most popular (by summed rating) movies the
    user has not rated, ``n_test_codes - 1`` pool draws per user under ``np.random.seed(seed)``.
    Returns the ``data`` dict ``ImplicitData`` takes; ``out`` also writes it the way the reference
    does (``np.savez(out, data=dict)``, a pickled dict: read it back with ``trust_pickle=True``)."""
    uid, mid, rating, ts = _read_dat(ratings_path, 4)
    user = np.asarray(uid, dtype=np.int32) - 1
    movie = _string_codes(mid)
    rating = np.asarray(rating, dtype=np.float32)
    ts = np.asarray(ts, dtype=np.int64)
    rank = _rank_first_within(user, ts)
    reverse = _rank_first_within(user, ts, descending=True)
    train, test = reverse > 0, reverse == 0

    np.random.seed(seed)
    user_movie = scs.csr_matrix((rating, (user, movie)))
    score = user_movie.sum(axis=0).repeat(user_movie.shape[0], axis=0)
    score[user_movie.nonzero()] = -1
    negative_samples = np.asarray(np.argsort(score, axis=1)[:, ::-1])[:, :pool]
    draws = np.random.randint(0, negative_samples.shape[1], (negative_samples.shape[0], n_test_codes - 1))
    negative_codes = np.take_along_axis(negative_samples, draws, 1)

    test_true = movie[test][:, None]
    rating_matrix = scs.csr_matrix((rating[train], (user[train], movie[train])))
    implicit_matrix = rating_matrix.copy()
    implicit_matrix[implicit_matrix.nonzero()] = 1
    timestamp_matrix = scs.csr_matrix((rank[train], (user[train], movie[train])))
    u_id, gender, age, occupation, _ = _read_dat(users_path, 5)
    user_infos = np.stack([np.asarray(u_id, dtype=np.int32) - 1, _string_codes(gender).astype(np.int32),
                           _string_codes(age).astype(np.int32), np.asarray(occupation, dtype=np.int32)], 1)
    data = {"rating_matrix": rating_matrix, "implicit_matrix": implicit_matrix, "timestamp_matrix": timestamp_matrix,
            "test_codes": np.hstack([test_true, negative_codes]), "test_true": test_true,
            "test_timestamp_rank": rank[test].astype(int), "negative_samples": negative_samples, "user_infos": user_infos}
    if out is not None:
        np.savez(out, data=data)
    return data


def load_ml1m_data(path, trust_pickle: bool = False):
    """Read the ``ml1m.npz`` written by the reference's own split (recsys/dataset.py:143-156).  That
    file stores a pickled dict, and unpickling runs code from the file: it is only read when the
    caller says so (``trust_pickle=True``) for a file they produced themselves."""
    if not trust_pickle:
        raise ValueError(f"{path}: the reference's ml1m.npz holds a pickled dict; pass trust_pickle=True "
                         "to load a file you wrote yourself")
    return np.load(path, allow_pickle=True)["data"].item()


def load_implicit_data(device="cpu", train_batch_size: int = 512, test_batch_size: int = 1024, path=None,
                       sampler: str = "device", trust_pickle: bool = False) -> ImplicitData:
    """``path`` to a user-built ml1m.npz (``split_data_loo(..., out=path)``), or ML-1M-shaped
    synthetic data when None."""
    data = load_ml1m_data(path, trust_pickle) if path is not None else synthetic_ml1m_data()
    return ImplicitData(data, device=device, train_batch_size=train_batch_size, test_batch_size=test_batch_size,
                        sampler=sampler)
