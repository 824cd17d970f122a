"""Functional CPU restatement of the reference's ten embedding models.

Every ``*_logits`` function takes ``P`` (a ``dict`` keyed exactly like the
reference module's ``state_dict()``) and the batch dict of
recsys/dataset.py:255-262 and returns ``[B, N]`` fp32 logits.  ``OracleModel``
wraps one of them with leaf tensors so that loss / backward / a torch optimizer
run exactly as in recsys/utils.py:96-103.

Test infrastructure: see ``oracle/__init__.py``.
"""
from __future__ import annotations

import math
from typing import Callable, Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- helpers
def _rows(table: Tensor, idx: Tensor) -> Tensor:
    return table.index_select(0, idx.reshape(-1)).reshape(*idx.shape, table.shape[1])


def _mf_dot(u: Tensor, v: Tensor) -> Tensor:
    # u [B, D], v [B, N, D] -> [B, N]   (recsys/models/matrix_factorization/vanilla_mf.py:29-31)
    return (u[:, None, :] * v).sum(-1)


def _mf_bias(P, users, items) -> Tensor:
    # recsys/models/matrix_factorization/bias_mf.py:16-18
    return P["bias"] + P["user_bias"][users][:, None] + P["item_bias"][items]


def _shared_indices(P, inputs, n_users, n_items, n_occ):
    # field offsets into the single shared table,
    # recsys/models/matrix_factorization/factorization_machine.py:33-38
    users = inputs["user_code"]
    item_idx = inputs["item_code"] + n_users
    ex_idx = torch.stack(
        [
            users,
            inputs["user_occupation"] + (n_users + n_items),
            inputs["item_timestamp_rank"] + (n_users + n_items + n_occ),
        ],
        dim=1,
    )
    return ex_idx, item_idx


def fm_second_order(ex: Tensor, item: Tensor) -> Tensor:
    """0.5 * sum_d[(sum_f e_f + v)^2 - (sum_f e_f^2 + v^2)]; ex [B,F,D], item [B,N,D] -> [B,N]
    (recsys/models/matrix_factorization/factorization_machine.py:52-63)."""
    s = ex.sum(1, keepdim=True) + item
    q = (ex * ex).sum(1, keepdim=True) + item * item
    return 0.5 * (s * s - q).sum(-1)


def _bn_tower(P, prefix: str, x: Tensor, train: bool, buffers_out: dict | None, dropout: float, linear=F.linear):
    """Linear -> BatchNorm1d -> ReLU -> Dropout, three times, then Linear(.,1)
    (recsys/models/neuralcf/widendeep.py:6-24; indices 0,1,4,5,8,9,12 in the Sequential).
    ``linear``: the Linear op (the bf16-operand towers of oracle.fields pass their own)."""
    for lin, bn in ((0, 1), (4, 5), (8, 9)):
        x = linear(x, P[f"{prefix}.{lin}.weight"], P[f"{prefix}.{lin}.bias"])
        rm, rv = P[f"{prefix}.{bn}.running_mean"], P[f"{prefix}.{bn}.running_var"]
        if train:
            rm, rv = rm.clone(), rv.clone()
        x = F.batch_norm(
            x, rm, rv, P[f"{prefix}.{bn}.weight"], P[f"{prefix}.{bn}.bias"],
            training=train, momentum=0.1, eps=1e-5,
        )
        if train and buffers_out is not None:
            buffers_out[f"{prefix}.{bn}.running_mean"] = rm
            buffers_out[f"{prefix}.{bn}.running_var"] = rv
            buffers_out[f"{prefix}.{bn}.num_batches_tracked"] = (
                P[f"{prefix}.{bn}.num_batches_tracked"] + 1
            )
        x = F.relu(x)
        x = F.dropout(x, dropout, training=train)
    return linear(x, P[f"{prefix}.12.weight"], P[f"{prefix}.12.bias"])


def _relu_tower(P, prefix: str, x: Tensor, n_linear: int, train: bool, dropout: float):
    """Linear/ReLU/Dropout stack with no activation after the last Linear
    (recsys/models/neuralcf/mlp.py:27-36: Linear sits at Sequential index 3*i)."""
    for i in range(n_linear):
        x = F.linear(x, P[f"{prefix}.{3 * i}.weight"], P[f"{prefix}.{3 * i}.bias"])
        if i + 1 < n_linear:
            x = F.dropout(F.relu(x), dropout, training=train)
    return x


def _count_linear(P, prefix: str) -> int:
    return sum(1 for k in P if k.startswith(prefix + ".") and k.endswith(".weight") and P[k].dim() == 2)


# --------------------------------------------------------------------------- MF family
def vanilla_mf_logits(P, inputs, **_):
    u = _rows(P["user_embedding.weight"], inputs["user_code"])
    v = _rows(P["item_embedding.weight"], inputs["item_code"])
    return _mf_dot(u, v)


def bias_mf_logits(P, inputs, **_):
    return _mf_bias(P, inputs["user_code"], inputs["item_code"]) + vanilla_mf_logits(P, inputs)


def side_features_mf_logits(P, inputs, **_):
    # recsys/models/matrix_factorization/side_features_mf.py:24-40
    u = _rows(P["user_embedding.weight"], inputs["user_code"]) + _rows(
        P["occupation_embedding.weight"], inputs["user_occupation"]
    )
    v = _rows(P["item_embedding.weight"], inputs["item_code"])
    return _mf_bias(P, inputs["user_code"], inputs["item_code"]) + _mf_dot(u, v)


def temporal_mf_logits(P, inputs, **_):
    # recsys/models/matrix_factorization/temporal_mf.py:34-54
    base = side_features_mf_logits(P, inputs)
    t = _rows(P["user_temporal_embedding.weight"], inputs["user_code"]) * _rows(
        P["temporal_embedding.weight"], inputs["item_timestamp_rank"]
    )
    return base + t.sum(-1, keepdim=True)


def factorization_machine_logits(P, inputs, *, n_users, n_items, n_occ, **_):
    ex_idx, item_idx = _shared_indices(P, inputs, n_users, n_items, n_occ)
    b = P["bias_embedding.weight"][:, 0]
    bias = b[ex_idx].sum(1, keepdim=True) + b[item_idx]
    w = P["feature_embedding.weight"]
    return bias + fm_second_order(_rows(w, ex_idx), _rows(w, item_idx))


# --------------------------------------------------------------------------- NCF family
def gmf_logits(P, inputs, *, pointwise=False, **_):
    # recsys/models/neuralcf/gmf.py:47-65
    u = _rows(P["user_embedding.weight"], inputs["user_code"])
    v = _rows(P["item_embedding.weight"], inputs["item_code"])
    if pointwise:
        return F.linear(u * v, P["linear.weight"], P["linear.bias"])
    prod = u[:, None, :] * v
    return F.linear(prod, P["linear.weight"], P["linear.bias"]).squeeze(-1)


def mlp_logits(P, inputs, *, train=False, dropout=0.0, pointwise=False, **_):
    # recsys/models/neuralcf/mlp.py:65-86 (concat order: user then item)
    u = _rows(P["user_embedding.weight"], inputs["user_code"])
    v = _rows(P["item_embedding.weight"], inputs["item_code"])
    n_lin = _count_linear(P, "mlp")
    if pointwise:
        return _relu_tower(P, "mlp", torch.cat([u, v], 1), n_lin, train, dropout)
    x = torch.cat([u[:, None, :].expand_as(v), v], -1)
    out = _relu_tower(P, "mlp", x.reshape(-1, x.shape[-1]), n_lin, train, dropout)
    return out.reshape(v.shape[0], -1)


def neumf_logits(P, inputs, *, train=False, dropout=0.0, pointwise=False, **_):
    # recsys/models/neuralcf/neumf.py:81-125.  The list-wise path feeds linear_final with
    # [gmf, mlp] (:104) and the point-wise path with [mlp, gmf] (:122); both are kept.
    ug = _rows(P["user_embedding_gmf.weight"], inputs["user_code"])
    vg = _rows(P["item_embedding_gmf.weight"], inputs["item_code"])
    um = _rows(P["user_embedding_mlp.weight"], inputs["user_code"])
    vm = _rows(P["item_embedding_mlp.weight"], inputs["item_code"])
    n_lin = _count_linear(P, "linear_mlp")
    if pointwise:
        g = F.linear(ug * vg, P["linear_gmf.weight"], P["linear_gmf.bias"])
        m = _relu_tower(P, "linear_mlp", torch.cat([um, vm], 1), n_lin, train, dropout)
        return F.linear(torch.cat([m, g], 1), P["linear_final.weight"], P["linear_final.bias"])
    b, n, d = vg.shape
    g = F.linear((ug[:, None, :] * vg).reshape(b * n, d), P["linear_gmf.weight"], P["linear_gmf.bias"])
    x = torch.cat([um[:, None, :].expand_as(vm), vm], -1).reshape(b * n, 2 * d)
    m = _relu_tower(P, "linear_mlp", x, n_lin, train, dropout)
    out = F.linear(torch.cat([g, m], 1), P["linear_final.weight"], P["linear_final.bias"])
    return out.reshape(b, n)


def _deep_input(w, ex_idx, item_idx):
    # [item row, user row, occupation row, timestamp row] per (b, n)
    # (recsys/models/neuralcf/widendeep.py:79-89)
    item = _rows(w, item_idx)
    b, n, d = item.shape
    ex = _rows(w, ex_idx).reshape(b, 1, -1).expand(b, n, 3 * d)
    return torch.cat([item, ex], -1).reshape(b * n, 4 * d), item


def widendeep_logits(P, inputs, *, n_users, n_items, n_occ, train=False, dropout=0.0, buffers_out=None, **_):
    ex_idx, item_idx = _shared_indices(P, inputs, n_users, n_items, n_occ)
    wide = P["wide_embedding.weight"][:, 0]
    wide_term = wide[ex_idx].sum(1, keepdim=True) + wide[item_idx] + P["wide_bias"]
    x, _ = _deep_input(P["deep_embedding.weight"], ex_idx, item_idx)
    deep = _bn_tower(P, "deep_mlp.mlp", x, train, buffers_out, dropout)
    return wide_term + deep.reshape(item_idx.shape)


def deepfm_logits(P, inputs, *, n_users, n_items, n_occ, train=False, dropout=0.0, buffers_out=None, **_):
    # recsys/models/neuralcf/deep_factorization_machine.py:53-107
    ex_idx, item_idx = _shared_indices(P, inputs, n_users, n_items, n_occ)
    b = P["bias_embedding.weight"][:, 0]
    bias = b[ex_idx].sum(1, keepdim=True) + b[item_idx]
    w = P["feature_embedding.weight"]
    x, item = _deep_input(w, ex_idx, item_idx)
    fm = fm_second_order(_rows(w, ex_idx), item)
    deep = _bn_tower(P, "deep_mlp.mlp", x, train, buffers_out, dropout)
    return bias + fm + deep.reshape(item_idx.shape)


LOGITS: Dict[str, Callable] = {
    "VanillaMF": vanilla_mf_logits,
    "BiasMF": bias_mf_logits,
    "SideFeaturesMF": side_features_mf_logits,
    "TemporalMF": temporal_mf_logits,
    "FactorizationMachine": factorization_machine_logits,
    "GMF": gmf_logits,
    "MLP": mlp_logits,
    "NeuMF": neumf_logits,
    "WidenDeep": widendeep_logits,
    "DeepFactorizationMachine": deepfm_logits,
}
POINTWISE = ("GMF", "MLP", "NeuMF")
DEFAULT_DROPOUT = 0.1


class _MeanBCE(torch.autograd.Function):
    """loss = mean(max(x, 0) - x y + log1p(exp(-|x|))), d loss / d x = (sigmoid(x) - y) / numel:
    the derivative aten's binary_cross_entropy_with_logits uses.  Spelled out because autograd
    through clamp / abs would give (1 - y) / numel at x == 0 exactly (a logit is exactly zero
    whenever every ReLU in front of a zero-bias output layer is off)."""

    @staticmethod
    def forward(ctx, x, y):
        ctx.save_for_backward(x, y)
        return (torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))).mean()

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        return (torch.sigmoid(x) - y) * (g / x.numel()), None


def bce_with_logits(logits: Tensor, labels: Tensor) -> Tensor:
    """Mean binary cross entropy; fp32 logits against fp64 labels promote to an fp64 loss,
    as F.binary_cross_entropy_with_logits does at e.g. vanilla_mf.py:35-39 with the labels
    of recsys/dataset.py:184."""
    x = logits.to(labels.dtype) if labels.dtype == torch.float64 else logits
    return _MeanBCE.apply(x, labels)


# --------------------------------------------------------------------------- init
def _xavier_uniform(out_f, in_f, gen):
    a = math.sqrt(6.0 / (in_f + out_f))
    return (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * a


def _default_linear(out_f, in_f, gen):
    bound = 1.0 / math.sqrt(in_f)
    w = (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound
    b = (torch.rand(out_f, generator=gen) * 2 - 1) * bound
    return w, b


def init_state(name: str, *, n_users, n_items, embedding_dim, n_occupations=0,
               max_timestamp_rank=0, seed=0) -> Dict[str, Tensor]:
    """Random state with the reference's parameter names, shapes and init distributions
    (N(0,1) embeddings for the MF / FM / W&D / DeepFM modules, N(0, 0.01) + xavier + zero
    bias for GMF / MLP / NeuMF, randn biases, wide_bias = 1).  Same distributions, not the
    same random stream as torch's module constructors."""
    g = torch.Generator().manual_seed(seed)
    d = embedding_dim
    n_feat = n_users + n_items + n_occupations + max_timestamp_rank
    rn = lambda *s, std=1.0: torch.randn(*s, generator=g) * std  # noqa: E731
    P: Dict[str, Tensor] = {}

    def bn_tower(prefix):
        dims = [4 * d, 4 * d, 2 * d, d]
        for k, (lin, bn) in enumerate(((0, 1), (4, 5), (8, 9))):
            w, b = _default_linear(dims[k + 1], dims[k], g)
            P[f"{prefix}.{lin}.weight"], P[f"{prefix}.{lin}.bias"] = w, b
            P[f"{prefix}.{bn}.weight"] = torch.ones(dims[k + 1])
            P[f"{prefix}.{bn}.bias"] = torch.zeros(dims[k + 1])
            P[f"{prefix}.{bn}.running_mean"] = torch.zeros(dims[k + 1])
            P[f"{prefix}.{bn}.running_var"] = torch.ones(dims[k + 1])
            P[f"{prefix}.{bn}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
        w, b = _default_linear(1, d, g)
        P[f"{prefix}.12.weight"], P[f"{prefix}.12.bias"] = w, b

    if name in ("VanillaMF", "BiasMF", "SideFeaturesMF", "TemporalMF"):
        P["user_embedding.weight"] = rn(n_users, d)
        P["item_embedding.weight"] = rn(n_items, d)
        if name != "VanillaMF":
            P["user_bias"], P["item_bias"], P["bias"] = rn(n_users), rn(n_items), rn(1)
        if name in ("SideFeaturesMF", "TemporalMF"):
            P["occupation_embedding.weight"] = rn(n_occupations, d)
        if name == "TemporalMF":
            P["user_temporal_embedding.weight"] = rn(n_users, d)
            P["temporal_embedding.weight"] = rn(max_timestamp_rank, d)
    elif name == "FactorizationMachine":
        P["feature_embedding.weight"] = rn(n_feat, d)
        P["bias_embedding.weight"] = rn(n_feat, 1)
    elif name == "GMF":
        P["user_embedding.weight"] = rn(n_users, d, std=0.01)
        P["item_embedding.weight"] = rn(n_items, d, std=0.01)
        P["linear.weight"], P["linear.bias"] = _xavier_uniform(1, d, g), torch.zeros(1)
    elif name == "MLP":
        P["user_embedding.weight"] = rn(n_users, d, std=0.01)
        P["item_embedding.weight"] = rn(n_items, d, std=0.01)
        dims = (2 * d, d, d // 2, 1)
        for i in range(3):
            P[f"mlp.{3 * i}.weight"] = _xavier_uniform(dims[i + 1], dims[i], g)
            P[f"mlp.{3 * i}.bias"] = torch.zeros(dims[i + 1])
    elif name == "NeuMF":
        for t in ("gmf", "mlp"):
            P[f"user_embedding_{t}.weight"] = rn(n_users, d, std=0.01)
            P[f"item_embedding_{t}.weight"] = rn(n_items, d, std=0.01)
        P["linear_gmf.weight"], P["linear_gmf.bias"] = _xavier_uniform(d // 2, d, g), torch.zeros(d // 2)
        dims = (2 * d, d, d // 2)
        for i in range(2):
            P[f"linear_mlp.{3 * i}.weight"] = _xavier_uniform(dims[i + 1], dims[i], g)
            P[f"linear_mlp.{3 * i}.bias"] = torch.zeros(dims[i + 1])
        P["linear_final.weight"], P["linear_final.bias"] = _xavier_uniform(1, d, g), torch.zeros(1)
    elif name == "WidenDeep":
        P["deep_embedding.weight"] = rn(n_feat, d)
        bn_tower("deep_mlp.mlp")
        P["wide_embedding.weight"] = rn(n_feat, 1)
        P["wide_bias"] = torch.ones(1)
    elif name == "DeepFactorizationMachine":
        P["feature_embedding.weight"] = rn(n_feat, d)
        P["bias_embedding.weight"] = rn(n_feat, 1)
        bn_tower("deep_mlp.mlp")
    else:
        raise KeyError(name)
    return P


# --------------------------------------------------------------------------- wrapper
class OracleModel:
    """Leaf-tensor wrapper so the functional restatement can be trained with torch optimizers
    following the step order of recsys/utils.py:96-103."""

    def __init__(self, name: str, state: Dict[str, Tensor], *, n_users, n_items,
                 n_occupations=0, dropout=None):
        self.name = name
        self.fn = LOGITS[name]
        self.dims = dict(n_users=n_users, n_items=n_items, n_occ=n_occupations)
        self.training = True
        self.dropout = DEFAULT_DROPOUT if dropout is None else dropout
        self.params: Dict[str, Tensor] = {}
        self.buffers: Dict[str, Tensor] = {}
        for k, v in state.items():
            t = torch.as_tensor(v).clone()
            is_buf = k.endswith(("running_mean", "running_var", "num_batches_tracked"))
            if is_buf:
                self.buffers[k] = t
            else:
                self.params[k] = t.requires_grad_(True)

    def parameters(self):
        return list(self.params.values())

    def named_parameters(self):
        return list(self.params.items())

    def state_dict(self):
        out = {k: v.detach().clone() for k, v in self.params.items()}
        out.update({k: v.clone() for k, v in self.buffers.items()})
        return out

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def forward(self, inputs, pointwise=False):
        P = {**self.params, **self.buffers}
        new_buf: dict = {}
        out = self.fn(P, inputs, train=self.training, dropout=self.dropout,
                      pointwise=pointwise, buffers_out=new_buf, **self.dims)
        for k, v in new_buf.items():
            self.buffers[k] = v.detach()
        return out

    __call__ = forward

    def loss(self, inputs, pointwise=False):
        return bce_with_logits(self.forward(inputs, pointwise), inputs["label"])


def reference_train_step(model, optimizer, inputs, clip_norm=1.0):
    """recsys/utils.py:96-106 for one batch; returns the loss as a Python float."""
    optimizer.zero_grad()
    loss = model.loss(inputs)
    loss.backward()
    if clip_norm is not None:
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip_norm)
    optimizer.step()
    return loss.detach().item()
