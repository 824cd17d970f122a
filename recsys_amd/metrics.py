"""NDCG@k / HR@k with the reference's definitions (recsys/metrics.py:4-38), computed on whatever
device the predictions live on instead of a per-row numpy loop on the host.

``true`` is [n, T] held-out item codes per user, ``pred`` [n, k] recommended codes in rank order
(numpy arrays or torch tensors)."""
from __future__ import annotations

import math

import numpy as np
import torch


def _t(x, device=None):
    if isinstance(x, torch.Tensor):
        return x
    return torch.as_tensor(np.asarray(x), device=device)


def ndcg_score(true, pred):
    pred = _t(pred)
    true = _t(true, pred.device).to(pred.device)
    n, k = pred.shape
    discount = torch.log2(torch.arange(2, k + 2, device=pred.device, dtype=torch.float64))
    rel = (pred[:, :, None] == true[:, None, :]).any(-1).to(torch.float64)   # isin(pred[i], true[i])
    irel = torch.zeros(k, dtype=torch.float64, device=pred.device)
    irel[: true.shape[1]] = 1
    dcg = (rel / discount).sum(1)
    idcg = (irel / discount).sum()
    return float((dcg / idcg).mean().item())


def hr_score(true, pred):
    pred = _t(pred)
    true = _t(true, pred.device).to(pred.device)
    rel = (true[:, :, None] == pred[:, None, :]).any(-1)                     # isin(true[i], pred[i])
    return float(rel.to(torch.float64).mean().item())


def compute_metrics(true, pred):
    k = pred.shape[1]
    ndcg = ndcg_score(true, pred)
    hr = hr_score(true, pred)
    print(f"metrics - ndcg@{k}: {ndcg:.4f}, hr@{k}: {hr:.4f}")
    return ndcg, hr
