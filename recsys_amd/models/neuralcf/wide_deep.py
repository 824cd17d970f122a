"""Wide & Deep and DeepFM over the shared 4-field table
(recsys/models/neuralcf/{widendeep,deep_factorization_machine}.py).

One rx_fm2_fwd launch gathers every row once and produces both the D=1 bias / wide sums
(+ the FM second-order term for DeepFM) and the [B*N, 4D] tower input laid out
[item, user, occupation, timestamp]; the tower's Linear layers run on the MFMA GEMM.
"""
from __future__ import annotations

from torch import nn
import torch

from ... import autograd as rxa
from ..base import HotPathModule, run_tower
from ..matrix_factorization.fm import SharedTableMixin


class MLP(nn.Module):
    """Linear -> BatchNorm1d -> ReLU -> Dropout per hidden width, then Linear(., 1); module
    indices inside ``mlp`` match widendeep.py:6-24 (0,1,2,3 | 4,5,6,7 | 8,9,10,11 | 12)."""

    def __init__(self, input_dim, hidden_dims, dropout=0.1):
        super().__init__()
        layers = []
        for width in hidden_dims:
            layers.extend([nn.Linear(input_dim, width), nn.BatchNorm1d(width), nn.ReLU(), nn.Dropout(p=dropout)])
            input_dim = width
        layers.append(nn.Linear(input_dim, 1))
        self.mlp = nn.Sequential(*layers)

    def forward(self, x):
        return run_tower(self.mlp, x)


class WidenDeep(HotPathModule, SharedTableMixin):
    _sparse_names = ["deep_embedding.weight", "wide_embedding.weight"]

    def __init__(self, n_users, n_items, n_occupations, max_timestamp_rank, embedding_dim):
        super().__init__()
        self._init_fields(n_users, n_items, n_occupations, max_timestamp_rank, embedding_dim)
        self.deep_embedding = nn.Embedding(self.n_features, embedding_dim)
        self.deep_mlp = MLP(embedding_dim * 4, [embedding_dim * 4, embedding_dim * 2, embedding_dim], dropout=0.1)
        self.wide_embedding = nn.Embedding(self.n_features, 1)
        self.wide_bias = nn.Parameter(torch.Tensor([1]), requires_grad=True)

    def forward(self, inputs):
        ex_idx, ex_base, items, item_base = self._field_indices(inputs)
        wide, deep_in = rxa.FM2.apply(ex_idx, ex_base, items, item_base, self.deep_embedding.weight,
                                      self.wide_embedding.weight, True, False)
        deep = self.deep_mlp(deep_in).view(items.shape[0], -1)
        return (wide + self.wide_bias) + deep


class DeepFactorizationMachine(HotPathModule, SharedTableMixin):
    _sparse_names = ["feature_embedding.weight", "bias_embedding.weight"]

    def __init__(self, n_users, n_items, n_occupations, max_timestamp_rank, embedding_dim):
        super().__init__()
        self._init_fields(n_users, n_items, n_occupations, max_timestamp_rank, embedding_dim)
        self.feature_embedding = nn.Embedding(self.n_features, embedding_dim)
        self.bias_embedding = nn.Embedding(self.n_features, 1)
        self.deep_mlp = MLP(embedding_dim * 4, [embedding_dim * 4, embedding_dim * 2, embedding_dim], dropout=0.1)

    def forward(self, inputs):
        ex_idx, ex_base, items, item_base = self._field_indices(inputs)
        fm, deep_in = rxa.FM2.apply(ex_idx, ex_base, items, item_base, self.feature_embedding.weight,
                                    self.bias_embedding.weight, True, True)
        deep = self.deep_mlp(deep_in).view(items.shape[0], -1)
        return fm + deep
