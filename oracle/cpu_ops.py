"""CPU stand-ins (plain torch) for the kernels the sharded-exchange code calls, with the names and
result conventions of ``recsys_amd.ops``.  They let the exchange logic (bucketise -> all-to-all ->
gather -> un-permute, and the gradient path back) run under gloo on machines without a GPU.

Test infrastructure: see ``oracle/__init__.py``.  Only tests inject this module as ``backend``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from recsys_amd import _lib


@dataclass
class Segment:
    table: torch.Tensor
    idx: torch.Tensor
    col: int
    div: int = 1
    stride: int = 1
    k: int = 0
    base: int = 0


@dataclass
class SortedIndex:
    n: int
    n_rows: int
    sorted_pos: torch.Tensor
    sorted_slot: torch.Tensor
    uniq_rows: torch.Tensor
    seg_off: torch.Tensor
    n_uniq: torch.Tensor
    inverse: Optional[torch.Tensor] = None


def bucketize_by_owner(idx, world, n_rows_global, n_valid=None):
    flat = idx.reshape(-1)
    n = flat.numel()
    m = int(n_valid) if n_valid is not None else n
    real = flat[:m]
    if ((real < 0) | (real >= n_rows_global)).any():
        raise IndexError("index out of range")
    owner = torch.cat([real % world, torch.full((n - m,), world, dtype=flat.dtype)])   # padding: spare bin, last
    perm = torch.sort(owner, stable=True)[1]
    local = torch.cat([real // world, torch.zeros(n - m, dtype=flat.dtype)])
    return torch.bincount(owner[:m], minlength=world)[:world], local[perm], perm.to(torch.int32)


def gather_concat(segs, out_rows, out_cols, out=None):
    out = torch.empty(out_rows, out_cols) if out is None else out
    i = torch.arange(out_rows)
    for s in segs:
        j = (i // s.div) * s.stride + s.k
        out[:, s.col:s.col + s.table.shape[1]] = s.table[s.idx.reshape(-1)[j] + s.base]
    return out


def unpermute_rows(rows, perm, n_valid=None, out=None):
    m = perm.numel() if n_valid is None else int(n_valid)
    if out is None:
        out = torch.empty_like(rows)
    out[perm[:m].long()] = rows[:m]
    return out


def permute_rows(rows, perm, n_valid=None):
    m = perm.numel() if n_valid is None else int(n_valid)
    out = torch.zeros((perm.numel(),) + tuple(rows.shape[1:]), dtype=rows.dtype)
    out[:m] = rows[perm[:m].long()]
    return out


def pack_by_owner(send_local, perm, counts, capacity, pad_rows=1):
    G, n = counts.numel(), send_local.numel()
    packed = (torch.arange(G * capacity) % capacity) % pad_rows
    src_of = torch.zeros(n, dtype=torch.int32)
    if int(counts.max()) > capacity:
        raise RuntimeError("a fixed-capacity exchange buffer overflowed")
    j = 0
    for g in range(G):
        c = int(counts[g])
        dest = g * capacity + torch.arange(c)
        packed[dest] = send_local[j:j + c]
        src_of[perm[j:j + c].long()] = dest.to(torch.int32)
        j += c
    return packed, src_of


def sort_dedup(idx, n_rows, want_inverse=False):
    flat = idx.reshape(-1)
    _, pos = torch.sort(flat, stable=True)
    uniq, inv, counts = torch.unique(flat, sorted=True, return_inverse=True, return_counts=True)
    seg = torch.zeros(uniq.numel() + 1, dtype=torch.int32)
    seg[1:] = counts.cumsum(0)
    return SortedIndex(flat.numel(), n_rows, pos.to(torch.int32), inv[pos].to(torch.int32), uniq, seg,
                       torch.tensor([uniq.numel()], dtype=torch.int32), inv.to(torch.int32))


def segreduce(occ, si, mode, target, *, state=None, lr=0.0, eps=0.0, scale=None, want_sq=False, occ_ld=None,
              dim=None, occ2=None, target2=None, state2=None):
    occ = occ.reshape(si.n, -1)
    nu = int(si.n_uniq)
    G = torch.zeros(nu, occ.shape[1]).index_add_(0, si.sorted_slot.long(), occ[si.sorted_pos.long()])
    rows = si.uniq_rows[:nu]
    sc = float(scale) if scale is not None else 1.0
    t = target.reshape(target.shape[0], -1)
    if mode == _lib.RX_SEG_DENSE:
        t[rows] = G
    elif mode == _lib.RX_SEG_COMPACT:
        t[:nu] = G
    elif mode == _lib.RX_SEG_SGD:
        t[rows] -= lr * sc * G
    elif mode == _lib.RX_SEG_ADAGRAD:
        state[rows] += (sc * G) ** 2
        t[rows] -= lr * sc * G / (state[rows].sqrt() + eps)
    else:
        state[rows] += ((sc * G) ** 2).mean(1)
        t[rows] -= lr * sc * G / (state[rows].sqrt()[:, None] + eps)
    sq = (G ** 2).sum()
    if target2 is not None:
        # companion scalar table addressed by the same rows
        G2 = torch.zeros(nu).index_add_(0, si.sorted_slot.long(), occ2.reshape(-1)[si.sorted_pos.long()])
        t2 = target2.reshape(-1)
        if mode == _lib.RX_SEG_DENSE:
            t2[rows] = G2
        elif mode == _lib.RX_SEG_COMPACT:
            t2[:nu] = G2
        elif mode == _lib.RX_SEG_SGD:
            t2[rows] -= lr * sc * G2
        else:
            s2 = state2.reshape(-1)
            s2[rows] += (sc * G2) ** 2
            t2[rows] -= lr * sc * G2 / (s2[rows].sqrt() + eps)
        sq = sq + (G2 ** 2).sum()
    return sq.reshape(1) if want_sq else None


def row_update(compact, si, mode, table, *, state=None, lr=0.0, eps=0.0, scale=None, compact2=None, table2=None,
               state2=None):
    nu = int(si.n_uniq)
    rows = si.uniq_rows[:nu]
    sc = float(scale) if scale is not None else 1.0
    for G, t, st in ((compact.reshape(si.n, -1)[:nu], table.reshape(table.shape[0], -1), state),
                     (None if compact2 is None else compact2.reshape(-1, 1)[:nu], None if table2 is None else table2.reshape(-1, 1),
                      None if state2 is None else state2.reshape(-1, 1))):
        if G is None:
            continue
        if mode == _lib.RX_SEG_SGD:
            t[rows] -= lr * sc * G
        elif mode == _lib.RX_SEG_ADAGRAD:
            st[rows] += (sc * G) ** 2
            t[rows] -= lr * sc * G / (st[rows].sqrt() + eps)
        else:
            raise NotImplementedError("row_update stand-in: SGD and Adagrad")


def sum_accumulate(x, acc):
    acc += x.double().sum()


def multi_sumsq_partials(grads):
    return torch.stack([(g.double() ** 2).sum() for g in grads])


def _clip(total, max_norm):
    tn = total.sqrt().float()
    return (max_norm / (tn + 1e-6)).clamp(max=1.0).reshape(1), tn.reshape(1)


def clip_scale(acc, max_norm):
    return _clip(acc.reshape(()), max_norm)


def clip_scale_parts(parts, acc, max_norm, parts64=None):
    total = torch.zeros((), dtype=torch.float64)
    for t in (parts, acc, parts64):
        if t is not None:
            total = total + t.double().sum()
    return _clip(total, max_norm)


def fm2_fwd(feat, bias, ex_idx, ex_base, item_idx, item_base, want_deep=False, second_order=True, deep_cols=None,
            want_logits=True, extra=None, want_tsum=False):
    ex = feat[ex_idx + ex_base[None, :]]
    item = feat[item_idx + item_base]
    b = bias[ex_idx + ex_base[None, :]].sum(1, keepdim=True) + bias[item_idx + item_base]
    s = ex.sum(1, keepdim=True) + item
    q = (ex * ex).sum(1, keepdim=True) + item * item
    logits = b + 0.5 * (s * s - q).sum(-1)
    deep = None
    if want_deep:
        B, F, D = ex.shape
        cols = deep_cols if deep_cols is not None else (1 + F) * D
        deep = torch.zeros(B, cols)
        deep[:, :D] = item[:, 0, :]
        deep[:, D:(1 + F) * D] = ex.reshape(B, F * D)
        if extra is not None:
            deep[:, (1 + F) * D:(1 + F) * D + extra.shape[1]] = extra
    return (logits, deep, s[:, 0, :].contiguous()) if want_tsum else (logits, deep)


def segreduce_fm(tsum, g, n_fields_ex, si, mode, target, *, table=None, state=None, target2=None, state2=None,
                 lr=0.0, eps=0.0, scale=None, want_sq=False, sq_out=None, deep=None):
    """G_r = sum_b g_b T_b - (sum_b g_b) e_r over the occurrences of row r; bias gradient sum_b g_b."""
    B = tsum.shape[0]
    split = B * n_fields_ex
    pos = si.sorted_pos.long()
    sample = torch.where(pos < split, pos // n_fields_ex, pos - split)
    nu = int(si.n_uniq)
    slot = si.sorted_slot.long()
    g = g.reshape(-1)
    gs = torch.zeros(nu).index_add_(0, slot, g[sample])
    G = torch.zeros(nu, tsum.shape[1]).index_add_(0, slot, g[sample][:, None] * tsum[sample])
    rows = si.uniq_rows[:nu]
    src = table if table is not None else target
    G = G - gs[:, None] * src[rows]
    if deep is not None:
        # the occurrence of extra field f adds its slice (1 + f) of the tower-input gradient, the item slice 0
        D = tsum.shape[1]
        field = torch.where(pos < split, 1 + pos % n_fields_ex, torch.zeros_like(pos))
        per_occ = deep.reshape(B, -1)[:, :(1 + n_fields_ex) * D].reshape(B, 1 + n_fields_ex, D)[sample, field]
        G = G + torch.zeros(nu, D).index_add_(0, slot, per_occ)
    if mode == _lib.RX_SEG_DENSE:
        target[rows] = G
        if target2 is not None:
            target2.reshape(-1)[rows] = gs
        return None
    if mode != _lib.RX_SEG_COMPACT:
        raise NotImplementedError("the CPU stand-in covers the compact and dense modes the sharded path uses")
    target[:nu] = G
    if target2 is not None:
        target2.reshape(-1)[:nu] = gs
    return None


def fm2_bwd(g, d_deep, feat, bias, ex_idx, ex_base, item_idx, item_base, second_order=True):
    ex = feat[ex_idx + ex_base[None, :]]            # [B, F, D]
    item = feat[item_idx + item_base]               # [B, N, D]
    B, F, D = ex.shape
    N = item.shape[1]
    g = g.reshape(B, N)
    S = ex.sum(1)
    Gs = g.sum(1)
    gv = (g[:, :, None] * item).sum(1)
    d_ex = Gs[:, None, None] * (S[:, None, :] - ex) + gv[:, None, :]
    d_item = g[:, :, None] * S[:, None, :]
    occ = torch.cat([d_ex.reshape(B * F, D), d_item.reshape(B * N, D)])
    bias_occ = torch.cat([Gs[:, None].expand(B, F).reshape(-1), g.reshape(-1)])
    return occ, bias_occ


def bce_loss(logits, labels):
    from .models import bce_with_logits
    return bce_with_logits(logits, labels)


def run_tower(seq, x, bf16=False):
    """The tower as plain torch modules (``bf16`` has no CPU counterpart here and is ignored)."""
    return seq(x)
