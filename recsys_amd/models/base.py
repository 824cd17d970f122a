"""Shared pieces of the drop-in modules."""
from __future__ import annotations

import os

from typing import Iterator, List

import torch
from torch import nn

from .. import autograd as rxa

Tensor = torch.Tensor


class HotPathModule(nn.Module):
    """nn.Module whose embedding lookups and interaction run on the HIP kernels.

    ``sparse_parameters()`` lists the embedding-table parameters (the ones a fused sparse
    optimizer updates row-wise); everything else in ``parameters()`` is dense."""

    _sparse_names: List[str] = []

    def sparse_parameters(self) -> Iterator[nn.Parameter]:
        named = dict(self.named_parameters())
        for n in self._sparse_names:
            if n in named:
                yield named[n]

    def _index_lists(self, inputs):
        """``[(index tensor, table height), ...]`` whose sort the backward will ask for."""
        return []

    def prepare(self, inputs, out=None):
        """Index preparation that depends on the batch only (sort, dedup), done ahead of the step —
        e.g. on a side stream while the previous step runs.  Returns ``inputs`` with the result
        attached; models without such work return ``inputs`` unchanged.  ``out``: a previously
        prepared batch whose buffers are overwritten in place."""
        lists = self._index_lists(inputs)
        if not lists:
            return inputs
        old = out.get("_rx_sorted") if out is not None else None
        res = dict(inputs)
        res["_rx_sorted"] = [rxa.presort(t, rows, out=old[k] if old else None) for k, (t, rows) in enumerate(lists)]
        return res

    def loss(self, inputs, *args, **kwargs):
        labels = inputs["label"]
        logits = self(inputs, *args, **kwargs)
        return rxa.bce_with_logits(logits, labels)


def as_index(t: Tensor, name: str) -> Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if t.dtype != torch.int64:
        raise TypeError(f"{name}: expected int64 indices, got {t.dtype}")
    return t.contiguous()


def _relu_chain(mods):
    """The Linear layers of a [Linear, ReLU] * n + [Linear] stack, or None for any other stack."""
    if len(mods) % 2 == 0:
        return None
    for i, m in enumerate(mods):
        if not isinstance(m, nn.Linear if i % 2 == 0 else nn.ReLU):
            return None
    return mods[0::2]


def run_tower(seq: nn.Sequential, x: Tensor, bf16: bool = False, x16: Tensor = None, into: Tensor = None) -> Tensor:
    """Evaluate a Linear/(BatchNorm1d)/ReLU/Dropout stack.  Linear (+ a directly following ReLU)
    goes through the MFMA GEMM with the ReLU fused into its epilogue; BatchNorm1d with the ReLU and
    Dropout behind it is one fused normalisation (rx_bn_*); anything else runs as the torch module.
    With ``bf16`` a pure Linear/ReLU stack of suitable widths runs as one node that keeps its
    activations in bf16 in HBM (autograd.MLPStack16; ``x16``: a bf16 copy of ``x`` if the producer
    made one); RX_NO_BF16_STORAGE=1 keeps fp32 storage.  ``into``: where a stack that ends in a plain
    Linear writes its output (a column block of the buffer the next layer reads, autograd.JoinColumns)."""
    mods = list(seq)
    if into is not None and (bf16 or not isinstance(mods[-1], nn.Linear)):
        raise ValueError("run_tower: `into` is for fp32 stacks that end in a Linear")
    chain = _relu_chain(mods) if bf16 and os.environ.get("RX_NO_BF16_STORAGE") != "1" else None
    if chain is not None and rxa.h16_shapes_ok(x, [m.weight for m in chain]):
        wb = []
        for m in chain:
            wb += [m.weight, m.bias]
        return rxa.MLPStack16.apply(x, x16, *wb)
    if bf16 and os.environ.get("RX_NO_BF16_STORAGE") != "1" and x.dim() == 2 and x.is_cuda:
        y = rxa.bn_tower16(seq, x)      # the reference's BatchNorm tower with bf16 operand storage
        if y is not None:
            return y
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear):
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            drop = mods[i + 2] if fuse and i + 2 < len(mods) and isinstance(mods[i + 2], nn.Dropout) else None
            if drop is not None and x.is_cuda and x.dim() == 2 and (not drop.training or drop.p == 0.0):
                x = rxa.linear(x, m.weight, m.bias, relu=True, bf16=bf16)        # an idle Dropout is the identity
                i += 3
            elif (drop is not None and x.is_cuda and x.dim() == 2 and 0.0 < drop.p < 1.0 and not drop.inplace
                  and os.environ.get("RX_NO_FUSED_DROPOUT") != "1"):
                # Linear + ReLU + Dropout as one node: torch's fused dropout forward, and one backward pass over
                # the pair's output instead of a masked scaling plus a ReLU backward (autograd.LinearAct)
                x = rxa.linear(x, m.weight, m.bias, relu=True, bf16=bf16, drop_p=float(drop.p))
                i += 3
            else:
                last = into is not None and i == len(mods) - 1
                x = rxa.linear(x, m.weight, m.bias, relu=fuse, bf16=bf16, into=into if last else None)
                i += 2 if fuse else 1
        elif isinstance(m, nn.BatchNorm1d) and x.dim() == 2 and m.affine:
            j = i + 1
            relu = j < len(mods) and isinstance(mods[j], nn.ReLU)
            j += 1 if relu else 0
            p = 0.0
            if j < len(mods) and isinstance(mods[j], nn.Dropout):
                p = float(mods[j].p)
                j += 1
            x = rxa.batch_norm_act(x, m, relu, p)
            i = j
        else:
            x = m(x)
            i += 1
    return x
