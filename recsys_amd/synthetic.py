"""Seeded synthetic inputs with the shapes of the benchmark configurations.

MovieLens-1M itself is not available (ratings.dat is absent from the reference checkout), so the
ML-1M configurations run on data with its shape: 6040 users x 3706 items, 994,169 training
positives, 4 sampled negatives per positive out of a 500-item per-user pool
(recsys/dataset.py:111-124,165-187).  The Criteo-shaped configurations use 26 sparse fields with
log-uniform cardinalities and Zipf-distributed indices plus 13 dense features.
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch

ML1M = dict(n_users=6040, n_items=3706, n_occupations=21, max_timestamp_rank=2316, n_train=994_169)


def criteo_field_dims(total_rows: int = 33_000_000, n_fields: int = 26, seed: int = 0) -> List[int]:
    """26 cardinalities, log-uniform between 10 and 10 M, rescaled so that they sum to about
    ``total_rows`` (small fields keep their size; only the large ones absorb the rescale)."""
    rng = np.random.RandomState(seed)
    dims = 10.0 ** rng.uniform(1.0, 7.0, size=n_fields)
    small = dims < 1e4
    scale = (total_rows - dims[small].sum()) / dims[~small].sum()
    dims[~small] *= scale
    return [max(int(round(x)), 2) for x in dims]


def zipf_indices(n: int, cardinality: int, alpha: float, gen: torch.Generator, device) -> torch.Tensor:
    """Power-law ids in [0, cardinality): inverse CDF of p(k) ~ k^-alpha on the continuous relaxation."""
    u = torch.rand(n, generator=gen, device=device, dtype=torch.float64)
    if abs(alpha - 1.0) < 1e-9:
        k = torch.exp(u * math.log(cardinality))
    else:
        a = 1.0 - alpha
        k = ((cardinality ** a - 1.0) * u + 1.0) ** (1.0 / a)
    return (k.floor().long() - 1).clamp_(0, cardinality - 1)


def criteo_batch(field_dims, batch: int, seed: int, device, n_dense: int = 13, alpha: float = 1.05,
                 label_dtype=torch.float64) -> Dict[str, torch.Tensor]:
    gen = torch.Generator(device=device).manual_seed(seed)
    cols = [zipf_indices(batch, c, alpha, gen, device) for c in field_dims]
    sparse = torch.stack(cols, dim=1).contiguous()
    dense = torch.log1p(torch.exp(torch.randn(batch, n_dense, generator=gen, device=device)))
    label = (torch.rand(batch, 1, generator=gen, device=device) < 0.25).to(label_dtype)
    return {"sparse": sparse, "dense": dense.float().contiguous(), "label": label}


def ml1m_batch(batch: int, seed: int, device, n_negatives: int = 4, **dims) -> Dict[str, torch.Tensor]:
    """One train batch with the dict layout / dtypes of recsys/dataset.py:255-262 on ML-1M-shaped
    id spaces: Zipf-like users and positives, negatives from the popular head."""
    d = {**ML1M, **dims}
    gen = torch.Generator(device=device).manual_seed(seed)
    users = zipf_indices(batch, d["n_users"], 0.6, gen, device)
    pos = zipf_indices(batch, d["n_items"], 1.0, gen, device)
    neg = torch.randint(0, min(500, d["n_items"]), (batch, n_negatives), generator=gen, device=device)
    items = torch.cat([pos[:, None], neg], dim=1).contiguous()
    label = torch.zeros(batch, n_negatives + 1, dtype=torch.float64, device=device)
    label[:, 0] = 1
    return {
        "user_code": users,
        "item_code": items,
        "user_occupation": torch.randint(0, d["n_occupations"], (batch,), generator=gen, device=device),
        "item_timestamp_rank": torch.randint(0, d["max_timestamp_rank"], (batch,), generator=gen, device=device),
        "label": label,
    }
