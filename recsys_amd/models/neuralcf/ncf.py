"""GMF / MLP / NeuMF on the fused kernels (recsys/models/neuralcf/{gmf,mlp,neumf}.py).

Names, init (N(0, 0.01) embeddings, xavier-uniform Linear weights, zero biases) and construction
order follow the reference, so state_dicts are interchangeable.  The list-wise NeuMF feeds
``linear_final`` with [gmf, mlp] and the point-wise path with [mlp, gmf], exactly as
neumf.py:104 and neumf.py:122 do.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from ... import autograd as rxa
from ..base import HotPathModule, as_index, run_tower


def _relu_stack(dims, dropout):
    mods = []
    for i, o in zip(dims[:-1], dims[1:]):
        mods.extend([nn.Linear(i, o), nn.ReLU(), nn.Dropout(p=dropout)])
    return nn.Sequential(*mods[:-2])  # no ReLU / Dropout after the last Linear


def _zero_linear_biases(module):
    for m in module.modules():
        if isinstance(m, nn.Linear) and m.bias is not None:
            m.bias.data.zero_()


def _tower_input(users, items, U, V):
    """[B*N, 2D] rows = [user row (broadcast over N), item row] (mlp.py:67-76)."""
    B = users.shape[0]
    N = items.shape[1] if items.dim() == 2 else 1
    D = U.shape[1]
    spec = [(0, users, 0, N, 1, 0, 0), (1, items, D, 1, 1, 0, 0)]
    return rxa.GatherConcat.apply(spec, B * N, 2 * D, U, V), B, N


class GMF(HotPathModule):
    _sparse_names = ["user_embedding.weight", "item_embedding.weight"]

    def __init__(self, n_users, n_items, embedding_dim):
        super().__init__()
        self.n_users, self.n_items, self.embedding_dim = n_users, n_items, embedding_dim
        self.user_embedding = nn.Embedding(n_users, embedding_dim)
        self.item_embedding = nn.Embedding(n_items, embedding_dim)
        self.linear = nn.Linear(embedding_dim, 1)
        self.init_weight()

    def init_weight(self):
        nn.init.normal_(self.user_embedding.weight, std=0.01)
        nn.init.normal_(self.item_embedding.weight, std=0.01)
        nn.init.xavier_uniform_(self.linear.weight)
        _zero_linear_biases(self)

    def _index_lists(self, inputs):
        if inputs.get("item_code") is None:
            return []
        return [(as_index(inputs["user_code"], "user_code"), self.n_users),
                (as_index(inputs["item_code"], "item_code"), self.n_items)]

    def forward(self, inputs, pointwise=False):
        users = as_index(inputs["user_code"], "user_code")
        items = inputs["item_code"]
        if items is None:
            items = torch.arange(self.n_items, device=users.device).expand(users.shape[0], self.n_items)
        items = as_index(items, "item_code")
        if pointwise and items.dim() != 1:
            raise ValueError("pointwise=True takes item_code of shape [B]")
        # Linear(D, 1) is folded into the score kernel; [B] items give [B, 1] as gmf.py:59-65 does
        return rxa.MFScore.apply(users, items, None, None, self.user_embedding.weight, self.item_embedding.weight,
                                 None, None, None, None, None, None, self.linear.weight, self.linear.bias)


    def loss(self, inputs, pointwise=False):
        """BCE loss (gmf.py:41-45); list-wise on the GPU the forward launch also forms the loss and dL/dlogit
        (autograd.MFLoss)."""
        labels, items, users = inputs["label"], inputs.get("item_code"), inputs["user_code"]
        if (not pointwise and os.environ.get("RX_NO_FUSED_MF_LOSS") != "1" and items is not None and isinstance(labels, torch.Tensor)
                and labels.is_cuda and users.is_cuda and labels.dtype in (torch.float32, torch.float64) and items.dim() == 2
                and tuple(labels.shape) == tuple(items.shape) and users.shape[0] > 0):
            return rxa.MFLoss.apply(as_index(users, "user_code"), as_index(items, "item_code"), None, None,
                                    self.user_embedding.weight, self.item_embedding.weight, None, None, None, None, None, None,
                                    self.linear.weight, self.linear.bias, labels)
        return super().loss(inputs, pointwise)


class MLP(HotPathModule):
    _sparse_names = ["user_embedding.weight", "item_embedding.weight"]

    def __init__(self, n_users, n_items, embedding_dim, mlp_dims=None, dropout=0.1):
        super().__init__()
        self.n_users, self.n_items, self.embedding_dim = n_users, n_items, embedding_dim
        self.user_embedding = nn.Embedding(n_users, embedding_dim)
        self.item_embedding = nn.Embedding(n_items, embedding_dim)
        self.mlp_dims = mlp_dims or (embedding_dim * 2, embedding_dim, embedding_dim // 2, 1)
        self.mlp = _relu_stack(self.mlp_dims, dropout)
        self.init_weight()

    def init_weight(self):
        nn.init.normal_(self.user_embedding.weight, std=0.01)
        nn.init.normal_(self.item_embedding.weight, std=0.01)
        for layer in self.mlp:
            if isinstance(layer, nn.Linear):
                nn.init.xavier_uniform_(layer.weight)
        _zero_linear_biases(self)

    def _index_lists(self, inputs):
        if inputs.get("item_code") is None:
            return []
        return [(as_index(inputs["user_code"], "user_code"), self.n_users),
                (as_index(inputs["item_code"], "item_code"), self.n_items)]

    def forward(self, inputs, pointwise=False):
        users = as_index(inputs["user_code"], "user_code")
        items = inputs["item_code"]
        if items is None:
            items = torch.arange(self.n_items, device=users.device).expand(users.shape[0], self.n_items)
        items = as_index(items, "item_code")
        if pointwise and items.dim() != 1:
            raise ValueError("pointwise=True takes item_code of shape [B]")
        x, B, N = _tower_input(users, items, self.user_embedding.weight, self.item_embedding.weight)
        out = run_tower(self.mlp, x)
        return out if pointwise else out.view(B, -1)


class NeuMF(HotPathModule):
    _sparse_names = ["user_embedding_gmf.weight", "item_embedding_gmf.weight",
                     "user_embedding_mlp.weight", "item_embedding_mlp.weight"]
    _branch_streams = True      # forward() forks a second stream (see graph.GraphedTrainStep._shared_branch)

    def __init__(self, n_users, n_items, embedding_dim, mlp_dims=None, dropout=0.1):
        super().__init__()
        self.n_users, self.n_items, self.embedding_dim = n_users, n_items, embedding_dim
        self.user_embedding_gmf = nn.Embedding(n_users, embedding_dim)
        self.item_embedding_gmf = nn.Embedding(n_items, embedding_dim)
        self.linear_gmf = nn.Linear(embedding_dim, embedding_dim // 2)
        self.user_embedding_mlp = nn.Embedding(n_users, embedding_dim)
        self.item_embedding_mlp = nn.Embedding(n_items, embedding_dim)
        self.mlp_dims = mlp_dims or (embedding_dim * 2, embedding_dim, embedding_dim // 2)
        self.linear_mlp = _relu_stack(self.mlp_dims, dropout)
        self.linear_final = nn.Linear(embedding_dim, 1)
        self.init_weight()

    def init_weight(self):
        for emb in (self.user_embedding_mlp, self.item_embedding_mlp, self.user_embedding_gmf, self.item_embedding_gmf):
            nn.init.normal_(emb.weight, std=0.01)
        for layer in self.linear_mlp:
            if isinstance(layer, nn.Linear):
                nn.init.xavier_uniform_(layer.weight)
        nn.init.xavier_uniform_(self.linear_gmf.weight)
        nn.init.xavier_uniform_(self.linear_final.weight)
        _zero_linear_biases(self)

    def _index_lists(self, inputs):
        if inputs.get("item_code") is None:
            return []
        return [(as_index(inputs["user_code"], "user_code"), self.n_users),
                (as_index(inputs["item_code"], "item_code"), self.n_items)]

    def _fused_step_ok(self, users, items, labels) -> bool:
        """The one-launch train step (rx_neumf_step) serves the list-wise loss of the default architecture at
        embedding_dim 64 on the GPU; anything else takes the per-layer path.  RX_NO_FUSED_NEUMF=1: A/B."""
        from ... import ops
        if os.environ.get("RX_NO_FUSED_NEUMF") == "1" or not torch.is_grad_enabled():
            return False
        mods = list(self.linear_mlp)
        return (users.is_cuda and items.dim() == 2 and labels.is_cuda and labels.numel() == items.numel()
                and labels.dtype in (torch.float32, torch.float64)
                and tuple(self.mlp_dims) == (2 * self.embedding_dim, self.embedding_dim, self.embedding_dim // 2)
                and len(mods) == 4 and isinstance(mods[0], nn.Linear) and isinstance(mods[1], nn.ReLU)
                and isinstance(mods[2], nn.Dropout) and not mods[2].inplace and isinstance(mods[3], nn.Linear)
                and all(m.bias is not None for m in (mods[0], mods[3], self.linear_gmf, self.linear_final))
                and all(p.is_contiguous() for p in self.parameters())
                and ops.neumf_step_supported(self.embedding_dim, items.shape[1]))

    def loss(self, inputs, pointwise=False):
        if not pointwise and inputs.get("item_code") is not None and "label" in inputs:
            users = as_index(inputs["user_code"], "user_code")
            items = as_index(inputs["item_code"], "item_code")
            labels = inputs["label"]
            if self._fused_step_ok(users, items, labels):
                drop = self.linear_mlp[2]
                p = float(drop.p) if (drop.training and drop.p > 0) else 0.0
                seed = None
                if p > 0.0:         # the dropout stream: a device counter the launch advances (graph replays draw anew)
                    seed = getattr(self, "_drop_seed", None)
                    if seed is None or seed.device != users.device:
                        if torch.cuda.is_current_stream_capturing():
                            raise RuntimeError("NeuMF: run one eager training step before recording (the dropout counter is made on first use)")
                        # derived from torch's seed without drawing from its generator
                        seed = self._drop_seed = torch.tensor([torch.initial_seed() % (2 ** 62)], dtype=torch.int64).to(users.device)
                return rxa.NeuMFLoss.apply(users, items, labels, p, seed,
                                           self.user_embedding_gmf.weight, self.item_embedding_gmf.weight,
                                           self.user_embedding_mlp.weight, self.item_embedding_mlp.weight,
                                           self.linear_mlp[0].weight, self.linear_mlp[0].bias,
                                           self.linear_mlp[3].weight, self.linear_mlp[3].bias,
                                           self.linear_gmf.weight, self.linear_gmf.bias,
                                           self.linear_final.weight, self.linear_final.bias)
        return super().loss(inputs, pointwise)

    def forward(self, inputs, pointwise=False):
        users = as_index(inputs["user_code"], "user_code")
        items = inputs["item_code"]
        if items is None:
            items = torch.arange(self.n_items, device=users.device).expand(users.shape[0], self.n_items)
        items = as_index(items, "item_code")
        if pointwise and items.dim() != 1:
            raise ValueError("pointwise=True takes item_code of shape [B]")
        # the GMF half shares nothing with the MLP tower until the concatenation: it runs on a second
        # stream, forward and (autograd keeps a node on its forward stream) backward
        x, B, N = _tower_input(users, items, self.user_embedding_mlp.weight, self.item_embedding_mlp.weight)
        wg, wm = self.linear_gmf.weight.shape[0], self.linear_mlp[-1].weight.shape[0]
        direct = (users.is_cuda and isinstance(self.linear_mlp[-1], nn.Linear) and wg % 4 == 0 and wm % 4 == 0
                  and os.environ.get("RX_NEUMF_CAT") != "1")
        if direct:
            # the two halves write their outputs straight into the column blocks of linear_final's input:
            # [gmf, mlp] list-wise, [mlp, gmf] point-wise (neumf.py:104, :122); no concatenation copy
            both = torch.empty(x.shape[0], wg + wm, dtype=torch.float32, device=users.device)
            blk_gmf, blk_mlp = (both[:, wm:], both[:, :wm]) if pointwise else (both[:, :wg], both[:, wg:])
        with rxa.Branch(users.device) as br:
            prod = rxa.GMFMul.apply(users, items, self.user_embedding_gmf.weight, self.item_embedding_gmf.weight)
            out_gmf = rxa.linear(prod, self.linear_gmf.weight, self.linear_gmf.bias, into=blk_gmf if direct else None)
            if direct and br.on:
                both.record_stream(br.side)
        out_mlp = run_tower(self.linear_mlp, x, into=blk_mlp if direct else None)
        br.join(out_gmf)
        if direct:
            both = rxa.JoinColumns.apply([both], *([out_mlp, out_gmf] if pointwise else [out_gmf, out_mlp]))
        else:
            both = torch.cat([out_mlp, out_gmf] if pointwise else [out_gmf, out_mlp], dim=1)
        out = rxa.linear(both, self.linear_final.weight, self.linear_final.bias)
        return out if pointwise else out.view(B, -1)
