"""Factorization machine over one shared table with field offsets, on rx_fm2_fwd / _bwd
(recsys/models/matrix_factorization/factorization_machine.py)."""
from __future__ import annotations

import torch
from torch import nn

from ... import autograd as rxa
from ..base import HotPathModule, as_index


class SharedTableMixin:
    """Index plumbing of the 4-field models: user | item | occupation | timestamp rank blocks of
    one table (factorization_machine.py:33-38)."""

    def _init_fields(self, n_users, n_items, n_occupations, max_timestamp_rank, embedding_dim):
        self.n_users, self.n_items = n_users, n_items
        self.n_occupations, self.max_timestamp_rank = n_occupations, max_timestamp_rank
        self.embedding_dim = embedding_dim
        self.n_features = n_users + n_items + n_occupations + max_timestamp_rank
        # first table row of the user / occupation / timestamp-rank blocks; a buffer (outside the state_dict) so that it
        # follows the module across devices and no forward pass has to create it (not possible while a graph is recorded)
        self.register_buffer("_ex_base", torch.tensor([0, n_users + n_items, n_users + n_items + n_occupations], dtype=torch.int64),
                             persistent=False)

    def _field_indices(self, inputs):
        users = as_index(inputs["user_code"], "user_code")
        items = as_index(inputs["item_code"], "item_code")
        occ = as_index(inputs["user_occupation"], "user_occupation")
        ts = as_index(inputs["item_timestamp_rank"], "item_timestamp_rank")
        ex_idx = torch.stack([users, occ, ts], dim=1).contiguous()
        base = self._ex_base
        if base.device != users.device:      # (the batch lives elsewhere than the module: fails later with a clear message)
            base = base.to(users.device)
        return ex_idx, base, items, self.n_users


class FactorizationMachine(HotPathModule, SharedTableMixin):
    _sparse_names = ["feature_embedding.weight", "bias_embedding.weight"]

    def __init__(self, n_users, n_items, n_occupations, max_timestamp_rank, embedding_dim):
        super().__init__()
        self._init_fields(n_users, n_items, n_occupations, max_timestamp_rank, embedding_dim)
        self.feature_embedding = nn.Embedding(self.n_features, embedding_dim)
        self.bias_embedding = nn.Embedding(self.n_features, 1)

    def forward(self, inputs):
        ex_idx, ex_base, items, item_base = self._field_indices(inputs)
        logits, _ = rxa.FM2.apply(ex_idx, ex_base, items, item_base, self.feature_embedding.weight,
                                  self.bias_embedding.weight, False, True)
        return logits
