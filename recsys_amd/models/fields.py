"""Multi-field models over Criteo-shaped inputs: 26 sparse fields sharing one row-offset table
(+ 13 dense features for the deep parts).  They have no counterpart in the reference; they
generalise its 4-field FactorizationMachine / DeepFactorizationMachine
(recsys/models/matrix_factorization/factorization_machine.py,
recsys/models/neuralcf/deep_factorization_machine.py) and add a DCN-v2 cross network.

Batch dict: ``sparse`` int64 [B, n_fields] (index inside each field), ``dense`` fp32
[B, n_dense], ``label`` [B, 1].  The last sparse field plays the reference's "item" role with
N = 1, so the same rx_fm2 kernels serve both shapes.  Tower input column order:
[last field, field 0 .. field n-2, dense].
"""
from __future__ import annotations

import os

from typing import Sequence

import torch
from torch import nn

from .. import autograd as rxa
from .base import HotPathModule, as_index, run_tower


class _FieldTable(HotPathModule):
    def _init_fields(self, field_dims: Sequence[int], embedding_dim: int):
        self.field_dims = [int(x) for x in field_dims]
        self.n_fields = len(self.field_dims)
        self.embedding_dim = embedding_dim
        self.n_features = sum(self.field_dims)
        offs = [0]
        for d in self.field_dims[:-1]:
            offs.append(offs[-1] + d)
        self._offsets = offs
        self._field_base = tuple(offs) + (self.n_features,)   # host row offsets of every field + table height
        # device copy of the non-item fields' offsets: a buffer outside the state_dict, so that it follows the module
        # across devices and no forward pass has to create it (not possible while a graph is recorded)
        self.register_buffer("_ex_base", torch.tensor(offs[:-1], dtype=torch.int64), persistent=False)

    # ---- whole-line rows -------------------------------------------------------------------------
    # MI355X moves memory in 128-byte lines.  A D = 16 row is 64 bytes, its bias entry 4 bytes in another
    # array: two lines per occurrence in the forward pass and, in the update, a second read-modify-write
    # whose 4-byte store is the expensive kind.  With ``line_rows`` the two parameters are views into one
    # [rows, 2 * D] buffer -- row in columns 0..D-1, bias entry in column D, zeros behind it -- so a row
    # and its bias share a line: the bias read rides on the row's, and the SGD update writes the line
    # whole.  Names, shapes, values and arithmetic of ``feature_embedding.weight`` [rows, D] and
    # ``bias_embedding.weight`` [rows, 1] do not change (state_dicts stay interchangeable; results are
    # bit-identical to the back-to-back layout); only their strides do, and the table takes twice the
    # memory.  Such a model trains through the field-grouped backward only (one item per sample,
    # batches of up to 65536): the general sort path takes back-to-back rows and refuses these.
    def _pack_lines(self):
        feat, bias = self.feature_embedding.weight, self.bias_embedding.weight
        D = self.embedding_dim
        if D % 4 or feat.dtype != torch.float32:
            raise ValueError("line_rows: fp32 rows of a multiple of 4 floats only")
        with torch.no_grad():
            buf = torch.zeros(self.n_features, 2 * D, dtype=torch.float32, device=feat.device)
            buf[:, :D].copy_(feat)
            buf[:, D].copy_(bias.reshape(-1))
        self.feature_embedding.weight = nn.Parameter(buf[:, :D], requires_grad=feat.requires_grad)
        self.bias_embedding.weight = nn.Parameter(buf[:, D:D + 1], requires_grad=bias.requires_grad)

    def lines_packed(self) -> bool:
        feat, bias = self.feature_embedding.weight, getattr(self, "bias_embedding", None)
        D = self.embedding_dim
        return (bias is not None and feat.stride(0) == 2 * D and bias.weight.stride(0) == 2 * D
                and bias.weight.data_ptr() == feat.data_ptr() + 4 * D)

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() rebuild every parameter on its own, back to back: pack again
        out = super()._apply(fn, *args, **kwargs)
        if getattr(self, "line_rows", False) and not self.lines_packed():
            self._pack_lines()
        return out

    def _split(self, inputs):
        sparse = as_index(inputs["sparse"], "sparse")
        if sparse.dim() != 2 or sparse.shape[1] != self.n_fields:
            raise ValueError(f"sparse: expected [B, {self.n_fields}], got {tuple(sparse.shape)}")
        base = self._ex_base
        if base.device != sparse.device:
            base = base.to(sparse.device)
        F = self.n_fields - 1
        return sparse[:, :F], base, sparse[:, F:], self._offsets[-1]

    def prepare(self, inputs, out=None):
        from .. import ops
        ex, base, item, item_base = self._split(inputs)
        feat = self.feature_embedding.weight
        old = out.get("_rx_presort") if out is not None else None
        si = rxa.index_prep(feat, ex, base, item, item_base, self._field_base, out=old.si if old else None)
        if out is not None and old is not None:
            return out
        prepared = dict(inputs)
        prepared["_rx_presort"] = rxa.Presort(si)
        return prepared


class FieldFM(_FieldTable):
    """logit = sum_f bias[f] + 0.5 * sum_d[(sum_f e_f)^2 - sum_f e_f^2]."""
    _sparse_names = ["feature_embedding.weight", "bias_embedding.weight"]

    def __init__(self, field_dims, embedding_dim, line_rows: bool = False):
        super().__init__()
        self._init_fields(field_dims, embedding_dim)
        self.feature_embedding = nn.Embedding(self.n_features, embedding_dim)
        self.bias_embedding = nn.Embedding(self.n_features, 1)
        self.line_rows = bool(line_rows)
        if self.line_rows:
            self._pack_lines()

    def forward(self, inputs):
        ex, base, item, item_base = self._split(inputs)
        logits, _ = rxa.FM2.apply(ex, base, item, item_base, self.feature_embedding.weight,
                                  self.bias_embedding.weight, False, True, None, True, None,
                                  inputs.get("_rx_presort"), self._field_base)
        return logits

    def loss(self, inputs):
        """BCE loss.  With field groups in use the logits, the loss and the per-sample backward
        records come out of one launch (autograd.FMLoss); otherwise the generic two-step path."""
        ex, base, item, item_base = self._split(inputs)
        feat = self.feature_embedding.weight
        labels = inputs["label"]
        if (feat.is_cuda and labels.is_cuda and labels.numel() == ex.shape[0] and labels.dtype in (torch.float32, torch.float64)
                and rxa.use_field_groups(feat, ex, item, self._field_base)):
            return rxa.FMLoss.apply(ex, base, item, item_base, feat, self.bias_embedding.weight, labels,
                                    inputs.get("_rx_presort"), self._field_base)
        return super().loss(inputs)


def _bn_stack(input_dim, hidden_dims, dropout):
    layers = []
    for width in hidden_dims:
        layers.extend([nn.Linear(input_dim, width), nn.BatchNorm1d(width), nn.ReLU(), nn.Dropout(p=dropout)])
        input_dim = width
    layers.append(nn.Linear(input_dim, 1))
    return nn.Sequential(*layers)


class FieldDeepFM(_FieldTable):
    """FieldFM + the reference's BatchNorm tower over [embeddings, dense]; hidden widths follow the
    reference rule [in, in/2, in/4] (deep_factorization_machine.py:47-51)."""
    _sparse_names = ["feature_embedding.weight", "bias_embedding.weight"]

    def __init__(self, field_dims, embedding_dim, n_dense=13, dropout=0.1, bf16=False):
        super().__init__()
        self._init_fields(field_dims, embedding_dim)
        self.n_dense, self.bf16 = n_dense, bf16   # bf16: tower GEMM operands rounded to bf16, fp32 accumulation
        self.feature_embedding = nn.Embedding(self.n_features, embedding_dim)
        self.bias_embedding = nn.Embedding(self.n_features, 1)
        d_in = self.n_fields * embedding_dim + n_dense
        self.pad = _tower_pad(d_in, n_dense)
        d_in += self.pad
        self.deep_mlp = _bn_stack(d_in, [d_in, d_in // 2, d_in // 4], dropout)

    def forward(self, inputs):
        ex, base, item, item_base = self._split(inputs)
        d_emb = self.n_fields * self.embedding_dim
        fm, x = rxa.FM2.apply(ex, base, item, item_base, self.feature_embedding.weight,
                              self.bias_embedding.weight, True, True, d_emb + self.n_dense + self.pad, True,
                              _dense_input(inputs, self.n_dense, ex.shape[0], self.pad), inputs.get("_rx_presort"),
                              self._field_base)
        return fm + run_tower(self.deep_mlp, x, bf16=self.bf16)


def _dense_input(inputs, n_dense, rows, pad=0):
    """The dense features, followed by ``pad`` zero columns (see ``_tower_pad``)."""
    if not n_dense:
        return None
    dense = inputs["dense"]
    if dense.dtype != torch.float32 or tuple(dense.shape) != (rows, n_dense):
        raise ValueError(f"dense: expected fp32 [{rows}, {n_dense}]")
    return torch.nn.functional.pad(dense, (0, pad)) if pad else dense


def _tower_pad(width, n_dense):
    """Zero columns appended to the tower input so that its width is a multiple of 16: rows of
    the activations and of the first weights are then whole 16-byte vectors and whole MFMA k-steps
    (Criteo shape: 26 * 16 + 13 = 429 -> 432).  The padding is inert: it multiplies weight columns
    that therefore receive zero gradient, and a DCN-v2 cross layer keeps a zero coordinate zero."""
    return (-width) % 16 if n_dense else 0


class DCNv2(_FieldTable):
    """x0 = [26 embeddings, dense]; ``n_cross`` layers x_{l+1} = x0 * (W_l x_l + b_l) + x_l; then
    MLP (1024, 512, 256) with ReLU and a final Linear(., 1).  GEMM operands are rounded to bf16 for
    the matrix cores (fp32 accumulation, fp32 master weights) when ``bf16`` is set."""
    _sparse_names = ["feature_embedding.weight"]

    def __init__(self, field_dims, embedding_dim, n_dense=13, n_cross=3, mlp_dims=(1024, 512, 256), bf16=True):
        super().__init__()
        self._init_fields(field_dims, embedding_dim)
        self.n_dense, self.bf16 = n_dense, bf16
        self.feature_embedding = nn.Embedding(self.n_features, embedding_dim)
        d = self.n_fields * embedding_dim + n_dense
        self.pad = _tower_pad(d, n_dense)
        d += self.pad
        self.input_dim = d
        self.cross = nn.ModuleList([nn.Linear(d, d) for _ in range(n_cross)])
        layers, i = [], d
        for w in mlp_dims:
            layers.extend([nn.Linear(i, w), nn.ReLU()])
            i = w
        layers.append(nn.Linear(i, 1))
        self.mlp = nn.Sequential(*layers)
        nn.init.normal_(self.feature_embedding.weight, std=0.05)

    def forward(self, inputs):
        ex, base, item, item_base = self._split(inputs)
        d_emb = self.n_fields * self.embedding_dim
        _, x0 = rxa.FM2.apply(ex, base, item, item_base, self.feature_embedding.weight, None, True, False,
                              d_emb + self.n_dense + self.pad, False,
                              _dense_input(inputs, self.n_dense, ex.shape[0], self.pad), inputs.get("_rx_presort"),
                              self._field_base)
        if (self.bf16 and len(self.cross) and os.environ.get("RX_NO_BF16_STORAGE") != "1"
                and rxa.h16_shapes_ok(x0, [m.weight for m in self.cross])):
            # operands of every product kept as bf16 copies in HBM (same rounding, half the bytes)
            wb = []
            for m in self.cross:
                wb += [m.weight, m.bias]
            x, x16 = rxa.CrossStack16.apply(x0, *wb)
            return run_tower(self.mlp, x, bf16=True, x16=x16)
        x = rxa.cross_stack(x0, self.cross, self.bf16) if len(self.cross) else x0
        return run_tower(self.mlp, x, bf16=self.bf16)
