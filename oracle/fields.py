"""CPU restatement of the multi-field models (FieldFM, FieldDeepFM, DCN-v2).

These models are not in the reference, so there are no reference goldens for them: parity is
UNPINNED beyond the fact that ``field_fm_logits`` with 3 + 1 fields and N = 1 is the reference's
FactorizationMachine formula (checked in tests/test_oracle_fields.py against the golden FM
fixture).  Parameter names match ``recsys_amd.models.fields``.

Test infrastructure: see ``oracle/__init__.py``.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .models import fm_second_order


def _offsets(field_dims):
    offs = [0]
    for d in field_dims[:-1]:
        offs.append(offs[-1] + d)
    return torch.tensor(offs, dtype=torch.int64)


def _rows(P, key, sparse, field_dims):
    idx = sparse + _offsets(field_dims)[None, :]
    w = P[key]
    return w[idx.reshape(-1)].reshape(sparse.shape[0], sparse.shape[1], w.shape[1]), idx


def field_fm_logits(P, inputs, field_dims, **_):
    e, idx = _rows(P, "feature_embedding.weight", inputs["sparse"], field_dims)
    bias = P["bias_embedding.weight"][:, 0][idx].sum(1, keepdim=True)
    return bias + fm_second_order(e[:, :-1], e[:, -1:])


def _tower_input(e, dense, width=None):
    # [last field, field 0..n-2, dense, zero columns up to the first layer's input width]
    x = torch.cat([e[:, -1], e[:, :-1].reshape(e.shape[0], -1)], 1)
    x = torch.cat([x, dense], 1) if dense is not None and dense.shape[1] else x
    if width is not None and width > x.shape[1]:
        x = F.pad(x, (0, width - x.shape[1]))
    return x


def field_deepfm_logits(P, inputs, field_dims, train=False, dropout=0.0, buffers_out=None, bf16=False, **_):
    from .models import _bn_tower
    e, idx = _rows(P, "feature_embedding.weight", inputs["sparse"], field_dims)
    bias = P["bias_embedding.weight"][:, 0][idx].sum(1, keepdim=True)
    fm = fm_second_order(e[:, :-1], e[:, -1:])
    x = _tower_input(e, inputs.get("dense"), P["deep_mlp.0.weight"].shape[1])
    linear = (lambda a, w, b: _linear(a, w, b, True)) if bf16 else F.linear
    return bias + fm + _bn_tower(P, "deep_mlp", x, train, buffers_out, dropout, linear=linear)


def _r16(t):
    """Round to bfloat16 (nearest even) and widen back: the value a matrix-core operand carries."""
    return t.to(torch.bfloat16).to(torch.float32)


class _LinearBf16Operands(torch.autograd.Function):
    """y = r(x) r(w)^T + b with r = round-to-bf16 and fp32 accumulation; the backward GEMMs round
    their operands the same way (dx = r(dy) r(w), dw = r(dy)^T r(x)), the bias gradient sums the
    unrounded dy.  This is the arithmetic of a bf16-operand tower with fp32 master weights, not the
    straight-through gradient autograd would derive from the rounding."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return F.linear(_r16(x), _r16(w), b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        d = _r16(dy)
        return d @ _r16(w), d.t() @ _r16(x), dy.sum(0)


def _linear(x, w, b, bf16):
    return _LinearBf16Operands.apply(x, w, b) if bf16 else F.linear(x, w, b)


def dcnv2_logits(P, inputs, field_dims, bf16=False, **_):
    e, _ = _rows(P, "feature_embedding.weight", inputs["sparse"], field_dims)
    x0 = _tower_input(e, inputs.get("dense"), P["cross.0.weight"].shape[1])
    x = x0
    l = 0
    while f"cross.{l}.weight" in P:
        x = x0 * _linear(x, P[f"cross.{l}.weight"], P[f"cross.{l}.bias"], bf16) + x
        l += 1
    i = 0
    while f"mlp.{i}.weight" in P:
        x = _linear(x, P[f"mlp.{i}.weight"], P[f"mlp.{i}.bias"], bf16)
        if f"mlp.{i + 2}.weight" in P:
            x = F.relu(x)
        i += 2
    return x


class FieldOracle:
    """Leaf-tensor wrapper (same role as oracle.models.OracleModel)."""

    FN = {"FieldFM": field_fm_logits, "FieldDeepFM": field_deepfm_logits, "DCNv2": dcnv2_logits}

    def __init__(self, name, state, field_dims, dropout=0.0, bf16=False):
        self.fn, self.field_dims, self.dropout, self.training = self.FN[name], list(field_dims), dropout, True
        self.bf16 = bf16          # DCNv2 / FieldDeepFM towers: GEMM operands rounded to bfloat16
        self.params, self.buffers = {}, {}
        for k, v in state.items():
            t = torch.as_tensor(v).detach().clone()
            if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                self.buffers[k] = t
            else:
                self.params[k] = t.requires_grad_(True)

    def parameters(self):
        return list(self.params.values())

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def state_dict(self):
        out = {k: v.detach().clone() for k, v in self.params.items()}
        out.update({k: v.clone() for k, v in self.buffers.items()})
        return out

    def __call__(self, inputs):
        new_buf = {}
        out = self.fn({**self.params, **self.buffers}, inputs, self.field_dims, train=self.training,
                      dropout=self.dropout, buffers_out=new_buf, bf16=self.bf16)
        for k, v in new_buf.items():
            self.buffers[k] = v.detach()
        return out

    def loss(self, inputs):
        from .models import bce_with_logits
        return bce_with_logits(self(inputs), inputs["label"])
