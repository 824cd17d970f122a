"""Matrix-factorization family on the fused rx_mf_score kernels.

Constructor signatures, attribute / parameter names, init distributions and construction order
follow recsys/models/matrix_factorization/{vanilla_mf,bias_mf,side_features_mf,temporal_mf}.py,
so ``state_dict()`` is interchangeable with the reference modules and the same torch seed
gives the same initial weights.
"""
from __future__ import annotations

import torch
from torch import nn

import os

from ... import autograd as rxa
from ..base import HotPathModule, as_index

_FUSED_LOSS = os.environ.get("RX_NO_FUSED_MF_LOSS") != "1"   # A/B timing: forward and loss as two launches


class VanillaMF(HotPathModule):
    _sparse_names = ["user_embedding.weight", "item_embedding.weight"]

    def __init__(self, n_users, n_items, embedding_dim):
        super().__init__()
        self.n_users, self.n_items, self.embedding_dim = n_users, n_items, embedding_dim
        self.user_embedding = nn.Embedding(n_users, embedding_dim)
        self.item_embedding = nn.Embedding(n_items, embedding_dim)

    # optional pieces, overridden by the subclasses
    def _extras(self, inputs):
        return dict(occs=None, temps=None, O=None, UT=None, T=None, user_bias=None, item_bias=None, bias=None)

    def _index_lists(self, inputs):
        if inputs.get("item_code") is None:
            return []
        lists = [(as_index(inputs["user_code"], "user_code"), self.n_users),
                 (as_index(inputs["item_code"], "item_code"), self.n_items)]
        e = self._extras(inputs)
        if e["occs"] is not None:
            lists.append((e["occs"], e["O"].shape[0]))
        if e["temps"] is not None:
            lists.append((e["temps"], e["T"].shape[0]))
        return lists

    def forward(self, inputs):
        users = as_index(inputs["user_code"], "user_code")
        items = inputs["item_code"]
        if items is None:
            # score every item (vanilla_mf.py:21-27): same kernel, N = n_items
            items = torch.arange(self.n_items, device=users.device).expand(users.shape[0], self.n_items)
        items = as_index(items, "item_code")
        e = self._extras(inputs)
        return rxa.MFScore.apply(users, items, e["occs"], e["temps"], self.user_embedding.weight,
                                 self.item_embedding.weight, e["O"], e["UT"], e["T"], e["user_bias"],
                                 e["item_bias"], e["bias"], None, None)


    def loss(self, inputs):
        """BCE loss (vanilla_mf.py:35-39).  On the GPU with one label per logit the logits, the loss and
        dL/dlogit come out of one launch (autograd.MFLoss); otherwise the generic two-step path."""
        labels, items = inputs["label"], inputs.get("item_code")
        users = inputs["user_code"]
        if (_FUSED_LOSS and items is not None and isinstance(labels, torch.Tensor) and labels.is_cuda and users.is_cuda
                and labels.dtype in (torch.float32, torch.float64) and items.dim() == 2 and tuple(labels.shape) == tuple(items.shape)
                and users.shape[0] > 0):
            e = self._extras(inputs)
            return rxa.MFLoss.apply(as_index(users, "user_code"), as_index(items, "item_code"), e["occs"], e["temps"],
                                    self.user_embedding.weight, self.item_embedding.weight, e["O"], e["UT"], e["T"],
                                    e["user_bias"], e["item_bias"], e["bias"], None, None, labels)
        return super().loss(inputs)


class BiasMF(VanillaMF):
    _sparse_names = VanillaMF._sparse_names + ["user_bias", "item_bias"]

    def __init__(self, n_users, n_items, embedding_dim):
        super().__init__(n_users, n_items, embedding_dim)
        self.user_bias = nn.Parameter(torch.randn(n_users), requires_grad=True)
        self.item_bias = nn.Parameter(torch.randn(n_items), requires_grad=True)
        self.bias = nn.Parameter(torch.randn(1), requires_grad=True)

    def _extras(self, inputs):
        e = super()._extras(inputs)
        e.update(user_bias=self.user_bias, item_bias=self.item_bias, bias=self.bias)
        return e

    def forward(self, inputs):
        if inputs["item_code"] is None:
            # the reference indexes item_bias[None] before its None check (bias_mf.py:16-20)
            raise TypeError("item_code is required")
        return super().forward(inputs)


class SideFeaturesMF(BiasMF):
    _sparse_names = BiasMF._sparse_names + ["occupation_embedding.weight"]

    def __init__(self, n_users, n_items, n_occupations, embedding_dim):
        super().__init__(n_users, n_items, embedding_dim)
        self.occupation_embedding = nn.Embedding(n_occupations, embedding_dim)

    def _extras(self, inputs):
        e = super()._extras(inputs)
        e.update(O=self.occupation_embedding.weight, occs=as_index(inputs["user_occupation"], "user_occupation"))
        return e


class TemporalMF(SideFeaturesMF):
    _sparse_names = SideFeaturesMF._sparse_names + ["user_temporal_embedding.weight", "temporal_embedding.weight"]

    def __init__(self, n_users, n_items, n_occupations, max_timestamp_rank, embedding_dim):
        super().__init__(n_users, n_items, n_occupations, embedding_dim)
        self.user_temporal_embedding = nn.Embedding(n_users, embedding_dim)
        self.temporal_embedding = nn.Embedding(max_timestamp_rank, embedding_dim)

    def _extras(self, inputs):
        e = super()._extras(inputs)
        e.update(UT=self.user_temporal_embedding.weight, T=self.temporal_embedding.weight,
                 temps=as_index(inputs["item_timestamp_rank"], "item_timestamp_rank"))
        return e
