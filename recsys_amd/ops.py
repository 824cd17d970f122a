"""Tensor-level wrappers over the C ABI (no autograd here; see ``autograd.py``).

Each wrapper validates its tensors (device, dtype, contiguity, shapes) on the host, hands raw
device pointers plus the current HIP stream to ``librecsys_hip.so`` and returns torch tensors.
PyTorch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import FMArgs, GatherSeg, MFArgs, check, lib

Tensor = torch.Tensor

# ------------------------------------------------------------------ index-error reporting
_INDEX_CHECK = "sync"  # "sync": raise right after the op; "deferred": raise in raise_if_index_error()
_err_flags = {}


def set_index_check(mode: str) -> None:
    """``"sync"`` (default): every op that consumes indices synchronises and raises IndexError at
    once, like nn.Embedding on the CPU.  ``"deferred"``: ops only set a device flag; call
    :func:`raise_if_index_error` (e.g. once per epoch) to check it without per-step syncs."""
    global _INDEX_CHECK
    if mode not in ("sync", "deferred"):
        raise ValueError(mode)
    _INDEX_CHECK = mode


def _err_flag(device: torch.device) -> Tensor:
    key = device.index if device.index is not None else torch.cuda.current_device()
    t = _err_flags.get(key)
    if t is None:
        _lib.init_device(key)      # first use of this device: the library's per-device set-up (outside any capture)
        t = torch.zeros(1, dtype=torch.int32, device=device)
        _err_flags[key] = t
    return t


def raise_if_index_error(device=None) -> None:
    for key, t in list(_err_flags.items()):
        if device is not None and torch.device(device).index not in (None, key):
            continue
        bits = int(t.item())
        if bits:
            t.zero_()
        if bits & _lib.RX_ERR_INDEX:
            raise IndexError("index out of range in a recsys_amd embedding lookup")
        if bits & _lib.RX_ERR_CAPACITY:
            raise RuntimeError("a fixed-capacity exchange buffer overflowed: raise the capacity of the sharded lookup")


def _after_index_op(device) -> None:
    if _INDEX_CHECK == "sync":
        raise_if_index_error(device)


# ------------------------------------------------------------------ small helpers
def _stream(t: Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _dev(t: Tensor) -> int:
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


def _p(t: Optional[Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _req(t: Tensor, dtype, name: str, ndim: Optional[int] = None) -> Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a tensor on the GPU, got device {t.device}")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dimensions, got shape {tuple(t.shape)}")
    return t


def _same_device(ref: Tensor, *ts: Optional[Tensor]) -> None:
    for t in ts:
        if t is not None and t.device != ref.device:
            raise ValueError(f"tensors on different devices: {ref.device} vs {t.device}")


def _table(t: Tensor, name: str) -> Tensor:
    return _req(t, torch.float32, name, 2)


def _row_strided(t: Tensor, name: str, cols: Optional[int] = None):
    """fp32 table [rows, cols] on the GPU whose rows may be spaced out (unit inner stride, row stride a
    multiple of 4 floats, 16-byte aligned): (tensor, row stride).  The layout of a table that keeps
    every row in a cache line of its own; ops that take only back-to-back rows go through ``_table``."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2):
        raise TypeError(f"{name}: expected an fp32 [rows, cols] tensor on the GPU")
    if cols is not None and t.shape[1] != cols:
        raise ValueError(f"{name}: expected {cols} columns, got {t.shape[1]}")
    if t.is_contiguous():
        return t, t.shape[1]
    if (t.shape[1] > 1 and t.stride(1) != 1) or t.stride(0) < t.shape[1]:
        raise ValueError(f"{name}: rows must have unit inner stride and must not overlap")
    return t, t.stride(0)


def _idx(t: Tensor, name: str) -> Tensor:
    return _req(t, torch.int64, name)


def _ws(nbytes: int, device) -> Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------ gather / assemble
@dataclass
class Segment:
    """One column block of an assembled tower input (see rx_gather_seg in recsys_hip.h)."""
    table: Tensor
    idx: Tensor
    col: int
    div: int = 1
    stride: int = 1
    k: int = 0
    base: int = 0
    bag: int = 1        # > 1: the block is the sum of the rows of `bag` consecutive index elements


def _segs_array(segs: Sequence[Segment]):
    arr = (GatherSeg * len(segs))()
    for i, s in enumerate(segs):
        _table(s.table, f"segment {i} table")
        _idx(s.idx, f"segment {i} idx")
        arr[i] = GatherSeg(s.table.data_ptr(), s.idx.data_ptr(), s.table.shape[0], s.base,
                           s.table.shape[1], s.col, s.div, s.stride, s.k, s.bag)
    return arr


def gather_concat(segs: Sequence[Segment], out_rows: int, out_cols: int,
                  out: Optional[Tensor] = None) -> Tensor:
    ref = segs[0].table
    for s in segs:
        _same_device(ref, s.table, s.idx)
        need = ((out_rows - 1) // s.div) * s.stride + s.k + max(s.bag, 1) if out_rows else 0
        if s.idx.numel() < need:
            raise ValueError("gather_concat: index tensor shorter than the rows it must serve")
    if out is None:
        out = torch.empty(out_rows, out_cols, dtype=torch.float32, device=ref.device)
    else:
        _req(out, torch.float32, "out", 2)
    arr = _segs_array(segs)
    check(lib.rx_gather_concat(arr, len(segs), _p(out), out_rows, out.shape[1], _p(_err_flag(ref.device)),
                               _dev(ref), _stream(ref)), "rx_gather_concat")
    _after_index_op(ref.device)
    return out


def gather_concat_bwd(dout: Tensor, col: int, dim: int, div: int, out: Optional[Tensor] = None) -> Tensor:
    """``out``: a contiguous [rows // div, dim] destination (e.g. a slice of a larger buffer)."""
    _req(dout, torch.float32, "dout", 2)
    occ = out if out is not None else torch.empty(dout.shape[0] // div, dim, dtype=torch.float32, device=dout.device)
    if out is not None:
        _req(out, torch.float32, "out", 2)
        if tuple(out.shape) != (dout.shape[0] // div, dim) or not out.is_contiguous():
            raise ValueError("gather_concat_bwd: bad `out`")
    check(lib.rx_gather_concat_bwd(_p(dout), dout.shape[0], dout.shape[1], col, dim, div, _p(occ),
                                   _dev(dout), _stream(dout)), "rx_gather_concat_bwd")
    return occ


# ------------------------------------------------------------------ MF-family score
def _mf_args(U, V, users, items, O=None, occs=None, UT=None, T=None, temps=None,
             user_bias=None, item_bias=None, bias=None, lin_w=None, lin_b=None) -> Tuple[MFArgs, int, int]:
    _table(U, "user table"); _table(V, "item table")
    _idx(users, "user_code"); _idx(items, "item_code")
    if users.dim() != 1:
        raise ValueError(f"user_code: expected shape [B], got {tuple(users.shape)}")
    B = users.shape[0]
    if items.dim() == 1:
        N = 1
    elif items.dim() == 2:
        N = items.shape[1]
    else:
        raise ValueError(f"item_code: expected [B] or [B, N], got {tuple(items.shape)}")
    if items.shape[0] != B:
        raise ValueError("user_code and item_code disagree on the batch size")
    dim = U.shape[1]
    if V.shape[1] != dim:
        raise ValueError("user and item tables disagree on the embedding width")
    _same_device(U, V, users, items, O, occs, UT, T, temps, user_bias, item_bias, bias, lin_w, lin_b)
    for t, nm, shp in ((O, "occupation table", None), (UT, "user temporal table", (U.shape[0], dim)),
                       (T, "temporal table", None)):
        if t is not None:
            _table(t, nm)
            if t.shape[1] != dim or (shp and tuple(t.shape) != shp):
                raise ValueError(f"{nm}: bad shape {tuple(t.shape)}")
    for t, nm in ((occs, "user_occupation"), (temps, "item_timestamp_rank")):
        if t is not None:
            _idx(t, nm)
            if tuple(t.shape) != (B,):
                raise ValueError(f"{nm}: expected shape [{B}], got {tuple(t.shape)}")
    for t, nm, n in ((user_bias, "user_bias", U.shape[0]), (item_bias, "item_bias", V.shape[0]),
                     (bias, "bias", 1), (lin_w, "lin_w", dim), (lin_b, "lin_b", 1)):
        if t is not None:
            _req(t, torch.float32, nm)
            if t.numel() != n:
                raise ValueError(f"{nm}: expected {n} elements, got {t.numel()}")
    a = MFArgs(U.data_ptr(), U.shape[0], V.data_ptr(), V.shape[0],
               0 if O is None else O.data_ptr(), 0 if O is None else O.shape[0],
               0 if UT is None else UT.data_ptr(), 0 if T is None else T.data_ptr(),
               0 if T is None else T.shape[0],
               0 if user_bias is None else user_bias.data_ptr(),
               0 if item_bias is None else item_bias.data_ptr(),
               0 if bias is None else bias.data_ptr(),
               0 if lin_w is None else lin_w.data_ptr(), 0 if lin_b is None else lin_b.data_ptr(),
               users.data_ptr(), items.data_ptr(),
               0 if occs is None else occs.data_ptr(), 0 if temps is None else temps.data_ptr(),
               B, N, dim)
    return a, B, N


def mf_score_fwd(U, V, users, items, **kw) -> Tensor:
    a, B, N = _mf_args(U, V, users, items, **kw)
    logits = torch.empty(B, N, dtype=torch.float32, device=U.device)
    check(lib.rx_mf_score_fwd(C.byref(a), _p(logits), _p(_err_flag(U.device)), _dev(U), _stream(U)),
          "rx_mf_score_fwd")
    _after_index_op(U.device)
    return logits


def mf_score_fwd_bce(U, V, users, items, labels: Tensor, want_logits: bool = False, defer_final: bool = False, **kw):
    """rx_mf_score_fwd_bce: logits, mean BCE loss (fp64) and dL/dlogit of the MF family / GMF from one forward
    launch.  Returns (loss fp64 scalar, dlogits [B, N] fp32, logits [B, N] or None).  ``defer_final``: the
    one-wavefront launch that adds the block partials is left out; the returned loss is written by the
    ``mf_score_bwd`` call of the backward pass instead, to which the caller hands the fourth return value
    (``mf_score_bwd(..., loss_final=)``)."""
    a, B, N = _mf_args(U, V, users, items, **kw)
    if labels.dtype not in (torch.float64, torch.float32):
        raise TypeError(f"label: expected float64 or float32, got {labels.dtype}")
    _req(labels, labels.dtype, "label")
    if labels.numel() != B * N:
        raise ValueError(f"label: expected {B * N} elements")
    if B == 0:
        raise ValueError("mf_score_fwd_bce: empty batch")
    _same_device(U, labels)
    dev = U.device
    dlogits = torch.empty(B, N, dtype=torch.float32, device=dev)
    logits = torch.empty(B, N, dtype=torch.float32, device=dev) if want_logits else None
    loss = torch.empty((), dtype=torch.float64, device=dev)
    need = int(lib.rx_mf_score_fwd_bce_workspace_bytes(B, U.shape[1]))
    # deferred: the partials have to outlive this call, so they get a buffer of their own (a few KB)
    ws = torch.empty(need, dtype=torch.uint8, device=dev) if defer_final else _ws(need, dev)
    check(lib.rx_mf_score_fwd_bce(C.byref(a), _p(labels), int(labels.dtype == torch.float64), _p(logits), _p(dlogits),
                                  _p(None if defer_final else loss), _p(ws), ws.numel(), _p(_err_flag(dev)), _dev(U), _stream(U)),
          "rx_mf_score_fwd_bce")
    _after_index_op(dev)
    if defer_final:
        return loss, dlogits, logits, (ws, int(lib.rx_mf_score_fwd_bce_partials(B, U.shape[1])), B * N, loss)
    return loss, dlogits, logits


def mf_score_bwd(g: Tensor, U, V, users, items, loss_final=None, **kw):
    """Returns (user_occ [B,D], item_occ [B*N,D], utemp_occ, temp_occ, gsum [B], linw_grad [D]).
    ``loss_final``: (partials, their count, number of logits, loss tensor) of mf_score_fwd_bce(defer_final=True);
    the launch then also writes that loss."""
    a, B, N = _mf_args(U, V, users, items, **kw)
    if loss_final is not None:
        parts, n_parts, count, loss_out = loss_final
        a.loss_parts, a.n_loss_parts, a.loss_count, a.loss_out = parts.data_ptr(), n_parts, count, loss_out.data_ptr()
    _req(g, torch.float32, "grad_logits")
    if g.numel() != B * N:
        raise ValueError("grad_logits: wrong number of elements")
    dim, dev = U.shape[1], U.device
    user_occ = torch.empty(B, dim, dtype=torch.float32, device=dev)
    item_occ = torch.empty(B * N, dim, dtype=torch.float32, device=dev)
    has_t = kw.get("UT") is not None
    utemp_occ = torch.empty(B, dim, dtype=torch.float32, device=dev) if has_t else None
    temp_occ = torch.empty(B, dim, dtype=torch.float32, device=dev) if has_t else None
    gsum = torch.empty(B, dtype=torch.float32, device=dev)
    has_w = kw.get("lin_w") is not None
    part = None
    if has_w:
        grid = int(lib.rx_mf_bwd_grid(B, dim))
        part = torch.empty(max(grid, 1), dim, dtype=torch.float32, device=dev)
    check(lib.rx_mf_score_bwd(C.byref(a), _p(g), _p(user_occ), _p(item_occ), _p(utemp_occ), _p(temp_occ),
                              _p(gsum), _p(part), _dev(U), _stream(U)), "rx_mf_score_bwd")
    linw_grad = None
    if has_w:
        linw_grad = torch.empty(dim, dtype=torch.float32, device=dev)
        check(lib.rx_colsum_partials(_p(part), part.shape[0] if B else 0, dim, _p(linw_grad), _dev(U),
                                     _stream(U)), "rx_colsum_partials")
    return user_occ, item_occ, utemp_occ, temp_occ, gsum, linw_grad


# ------------------------------------------------------------------ GMF product
def gmf_mul_fwd(U, V, users, items, out: Optional[Tensor] = None, col: int = 0) -> Tensor:
    _table(U, "user table"); _table(V, "item table"); _idx(users, "user_code"); _idx(items, "item_code")
    _same_device(U, V, users, items)
    B = users.shape[0]
    N = items.shape[1] if items.dim() == 2 else 1
    dim = U.shape[1]
    if out is None:
        out = torch.empty(B * N, dim, dtype=torch.float32, device=U.device)
    _req(out, torch.float32, "out", 2)
    check(lib.rx_gmf_mul_fwd(_p(U), U.shape[0], _p(V), V.shape[0], _p(users), _p(items), B, N, dim, _p(out),
                             out.shape[1], col, _p(_err_flag(U.device)), _dev(U), _stream(U)), "rx_gmf_mul_fwd")
    _after_index_op(U.device)
    return out


def gmf_mul_bwd(dout: Tensor, U, V, users, items, col: int = 0):
    _req(dout, torch.float32, "dout", 2)
    B = users.shape[0]
    N = items.shape[1] if items.dim() == 2 else 1
    dim = U.shape[1]
    user_occ = torch.empty(B, dim, dtype=torch.float32, device=U.device)
    item_occ = torch.empty(B * N, dim, dtype=torch.float32, device=U.device)
    check(lib.rx_gmf_mul_bwd(_p(U), U.shape[0], _p(V), V.shape[0], _p(users), _p(items), B, N, dim, _p(dout),
                             dout.shape[1], col, _p(user_occ), _p(item_occ), _dev(U), _stream(U)), "rx_gmf_mul_bwd")
    return user_occ, item_occ


# ------------------------------------------------------------------ FM second order
def _idx_matrix(t: Tensor, name: str):
    """int64 [B, C] (or [B]) index matrix with unit inner stride; returns (tensor, cols, row stride).
    A column block of a wider matrix (e.g. ``sparse[:, :25]``) is accepted as is."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a tensor on the GPU, got device {t.device}")
    if t.dtype != torch.int64:
        raise TypeError(f"{name}: expected int64 indices, got {t.dtype}")
    if t.dim() == 1:
        return t, 1, (t.stride(0) if t.shape[0] > 1 else 1)
    if t.dim() != 2:
        raise ValueError(f"{name}: expected [B] or [B, C], got {tuple(t.shape)}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise ValueError(f"{name}: inner stride must be 1")
    ld = t.stride(0) if t.shape[0] > 1 else t.shape[1]
    return t, t.shape[1], max(ld, t.shape[1])


def _fm_args(feat, bias, ex_idx, ex_base, item_idx, item_base: int, second_order: bool = True,
             deep_ld: Optional[int] = None):
    feat, feat_ld = _row_strided(feat, "feature table")
    if feat_ld != feat.shape[1] and (feat_ld % 4 or feat.data_ptr() % 16):
        raise ValueError("feature table: spaced-out rows need a row stride that is a multiple of 4 floats and 16-byte alignment")
    _idx(ex_base, "ex_base")
    ex_idx, F, ex_ld = _idx_matrix(ex_idx, "ex_idx")
    item_idx, N, item_ld = _idx_matrix(item_idx, "item_idx")
    if ex_idx.dim() != 2:
        raise ValueError(f"ex_idx: expected [B, F], got {tuple(ex_idx.shape)}")
    B = ex_idx.shape[0]
    if ex_base.numel() != F:
        raise ValueError("ex_base: one offset per field expected")
    if item_idx.shape[0] != B:
        raise ValueError("ex_idx and item_idx disagree on the batch size")
    bias_ld = 1
    if bias is not None:
        if bias.dim() == 2 and bias.shape[1] == 1:
            bias = bias[:, 0]
        if not (bias.is_cuda and bias.dtype == torch.float32 and bias.dim() == 1):
            raise TypeError("bias table: expected an fp32 vector (or [rows, 1]) on the GPU")
        if bias.numel() != feat.shape[0]:
            raise ValueError("bias table: one entry per feature row expected")
        bias_ld = bias.stride(0) if bias.numel() > 1 else 1
        if bias_ld < 1:
            raise ValueError("bias table: bad stride")
    _same_device(feat, bias, ex_idx, ex_base, item_idx)
    dim = feat.shape[1]
    if deep_ld is None:
        deep_ld = (1 + F) * dim
    a = FMArgs(feat.data_ptr(), 0 if bias is None else bias.data_ptr(), feat.shape[0], ex_idx.data_ptr(),
               ex_base.data_ptr(), item_idx.data_ptr(), item_base, ex_ld, item_ld, deep_ld, B, F, N, dim,
               0 if second_order else _lib.RX_FM_NO_SECOND_ORDER, feat_ld, bias_ld)
    return a, B, F, N


def fm2_fwd(feat, bias, ex_idx, ex_base, item_idx, item_base: int, want_deep: bool = False,
            second_order: bool = True, deep_cols: Optional[int] = None, want_logits: bool = True,
            extra: Optional[Tensor] = None, want_tsum: bool = False):
    """``deep_cols`` > (1+F)*dim allocates a wider tower-input buffer whose spare columns receive
    ``extra`` (dense features, fp32 [B*N, deep_cols - (1+F)*dim]); the buffer's row stride is
    rounded up to a multiple of 4 floats."""
    dim = feat.shape[1]
    F = ex_idx.shape[1]
    deep_ld = None
    if want_deep:
        cols = max(deep_cols or 0, (1 + F) * dim)
        deep_ld = (cols + 3) // 4 * 4
    a, B, F, N = _fm_args(feat, bias, ex_idx, ex_base, item_idx, item_base, second_order, deep_ld)
    logits = torch.empty(B, N, dtype=torch.float32, device=feat.device) if want_logits else None
    deep = None
    if want_deep:
        buf = torch.empty(B * N, deep_ld, dtype=torch.float32, device=feat.device)
        deep = buf[:, :cols] if cols != deep_ld else buf
        if cols > (1 + F) * dim:
            if extra is None or tuple(extra.shape) != (B * N, cols - (1 + F) * dim) or extra.dtype != torch.float32:
                raise ValueError("extra: expected fp32 [B*N, deep_cols - (1+F)*dim]")
            buf[:, (1 + F) * dim:cols].copy_(extra)
    tsum = torch.empty(B, dim, dtype=torch.float32, device=feat.device) if want_tsum else None
    check(lib.rx_fm2_fwd(C.byref(a), _p(logits), _p(deep), _p(tsum), _p(_err_flag(feat.device)), _dev(feat),
                         _stream(feat)), "rx_fm2_fwd")
    _after_index_op(feat.device)
    return (logits, deep, tsum) if want_tsum else (logits, deep)


def fm2_bwd(g, d_deep, feat, bias, ex_idx, ex_base, item_idx, item_base: int, second_order: bool = True):
    """Returns (occ [(B*F+B*N), D], bias_occ [(B*F+B*N)] or None).  ``d_deep`` may be a column
    block (unit inner stride) of a wider gradient matrix."""
    deep_ld = None
    if d_deep is not None:
        d_deep, deep_ld = _mat(d_deep, "grad_deep")
        if feat.shape[1] % 4 == 0 and (deep_ld % 4 != 0 or d_deep.data_ptr() % 16 != 0):
            # 16-byte row accesses need an aligned row stride: re-home the gradient once
            padded = torch.empty(d_deep.shape[0], (d_deep.shape[1] + 3) // 4 * 4, dtype=torch.float32,
                                 device=d_deep.device)
            padded[:, :d_deep.shape[1]].copy_(d_deep)
            d_deep, deep_ld = padded, padded.shape[1]
    a, B, F, N = _fm_args(feat, bias, ex_idx, ex_base, item_idx, item_base, second_order, deep_ld)
    dim, dev = feat.shape[1], feat.device
    if g is None:
        g = torch.zeros(B * N, dtype=torch.float32, device=dev)
    _req(g, torch.float32, "grad_logits")
    if g.numel() != B * N:
        raise ValueError("grad_logits: wrong number of elements")
    if d_deep is not None and (d_deep.shape[0] != B * N or d_deep.shape[1] < (1 + F) * dim):
        raise ValueError("grad_deep: wrong shape")
    n = B * (F + N)
    occ = torch.empty(n, dim, dtype=torch.float32, device=dev)
    bias_occ = torch.empty(n, dtype=torch.float32, device=dev) if bias is not None else None
    check(lib.rx_fm2_bwd(C.byref(a), _p(g), _p(d_deep), _p(occ), _p(bias_occ), _dev(feat), _stream(feat)),
          "rx_fm2_bwd")
    return occ, bias_occ


def wall_clock(slot: Tensor) -> None:
    """Writes the device wall clock (10 ns ticks) into the one-element int64 tensor ``slot`` in
    stream order (rx_wall_clock); usable inside a captured graph."""
    _req(slot, torch.int64, "slot")
    check(lib.rx_wall_clock(_p(slot), _dev(slot), _stream(slot)), "rx_wall_clock")


def fm2_flat_indices(feat, ex_idx, ex_base, item_idx, item_base: int, out: Optional[Tensor] = None) -> Tensor:
    a, B, F, N = _fm_args(feat, None, ex_idx, ex_base, item_idx, item_base)
    flat = out if out is not None else torch.empty(B * (F + N), dtype=torch.int64, device=feat.device)
    if flat.numel() != B * (F + N) or flat.dtype != torch.int64:
        raise ValueError("fm2_flat_indices: bad `out`")
    check(lib.rx_fm2_flat_indices(C.byref(a), _p(flat), _dev(feat), _stream(feat)), "rx_fm2_flat_indices")
    return flat


# ------------------------------------------------------------------ BCE
def bce_with_logits(logits: Tensor, labels: Tensor, want_grad: bool = True):
    """Returns (loss fp64 scalar tensor, dlogits fp32 like logits or None)."""
    _req(logits, torch.float32, "logits")
    if labels.dtype not in (torch.float64, torch.float32):
        raise TypeError(f"label: expected float64 or float32, got {labels.dtype}")
    _req(labels, labels.dtype, "label")
    if labels.shape != logits.shape:
        raise ValueError(f"label shape {tuple(labels.shape)} != logits shape {tuple(logits.shape)}")
    _same_device(logits, labels)
    M = logits.numel()
    loss = torch.empty((), dtype=torch.float64, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    nb = int(lib.rx_bce_workspace_bytes(M))
    ws = _ws(nb, logits.device)
    check(lib.rx_bce_fwd(_p(logits), _p(labels), int(labels.dtype == torch.float64), M, _p(loss), _p(dl), _p(ws),
                         ws.numel(), _dev(logits), _stream(logits)), "rx_bce_fwd")
    return loss, dl


# ------------------------------------------------------------------ sort / dedup / segmented reduce
@dataclass
class SortedIndex:
    """Result of rx_sort_dedup for one index list (device tensors; n_uniq stays on the device)."""
    n: int
    n_rows: int
    sorted_pos: Tensor   # int32 [n]
    sorted_slot: Tensor  # int32 [n]  index into uniq_rows per sorted position
    uniq_rows: Tensor    # int64 [n]  (first n_uniq valid)
    seg_off: Tensor      # int32 [n+1]
    n_uniq: Tensor       # int32 [1]
    inverse: Optional[Tensor] = None


def sort_dedup(idx: Tensor, n_rows: int, want_inverse: bool = False, out: Optional[SortedIndex] = None) -> SortedIndex:
    """``out``: a SortedIndex of the same length whose tensors are overwritten in place (static
    buffers for graph capture / pipelining)."""
    _idx(idx, "indices")
    flat = idx.reshape(-1)
    n, dev = flat.numel(), idx.device
    if out is not None:
        if out.n != n or out.n_rows != n_rows:
            raise ValueError("sort_dedup: `out` was made for a different index list")
        sorted_pos, sorted_slot, uniq, seg, nu, inv = (out.sorted_pos, out.sorted_slot, out.uniq_rows, out.seg_off,
                                                        out.n_uniq, out.inverse)
    else:
        sorted_pos = torch.empty(n, dtype=torch.int32, device=dev)
        sorted_slot = torch.empty(n, dtype=torch.int32, device=dev)
        uniq = torch.empty(n, dtype=torch.int64, device=dev)
        seg = torch.empty(n + 1, dtype=torch.int32, device=dev)
        nu = torch.empty(1, dtype=torch.int32, device=dev)
        inv = torch.empty(n, dtype=torch.int32, device=dev) if want_inverse else None
    ws = _ws(int(lib.rx_sort_dedup_workspace_bytes(n)), dev)
    check(lib.rx_sort_dedup(_p(flat), n, n_rows, _p(sorted_pos), _p(sorted_slot), _p(uniq), _p(seg), _p(nu),
                            _p(inv), _p(ws), ws.numel(), _p(_err_flag(dev)), _dev(idx), _stream(idx)), "rx_sort_dedup")
    return out if out is not None else SortedIndex(n, n_rows, sorted_pos, sorted_slot, uniq, seg, nu, inv)


def fm_sort_dedup(feat, ex_idx, ex_base, item_idx, item_base: int, out: Optional[SortedIndex] = None) -> SortedIndex:
    """sort_dedup(fm2_flat_indices(...)) without materialising the flat list (rx_sort_dedup_fm)."""
    a, B, F, N = _fm_args(feat, None, ex_idx, ex_base, item_idx, item_base)
    n, n_rows, dev = B * (F + N), feat.shape[0], feat.device
    if out is not None:
        if out.n != n or out.n_rows != n_rows:
            raise ValueError("fm_sort_dedup: `out` was made for a different batch shape")
        sorted_pos, sorted_slot, uniq, seg, nu = out.sorted_pos, out.sorted_slot, out.uniq_rows, out.seg_off, out.n_uniq
    else:
        sorted_pos = torch.empty(n, dtype=torch.int32, device=dev)
        sorted_slot = torch.empty(n, dtype=torch.int32, device=dev)
        uniq = torch.empty(n, dtype=torch.int64, device=dev)
        seg = torch.empty(n + 1, dtype=torch.int32, device=dev)
        nu = torch.empty(1, dtype=torch.int32, device=dev)
    ws = _ws(int(lib.rx_sort_dedup_workspace_bytes(n)), dev)
    check(lib.rx_sort_dedup_fm(C.byref(a), _p(sorted_pos), _p(sorted_slot), _p(uniq), _p(seg), _p(nu), None,
                               _p(ws), ws.numel(), _p(_err_flag(dev)), _dev(feat), _stream(feat)), "rx_sort_dedup_fm")
    return out if out is not None else SortedIndex(n, n_rows, sorted_pos, sorted_slot, uniq, seg, nu, None)


def segreduce_grid(n: int, dim: int) -> int:
    """Number of per-block partials a reduce over n occurrences of width dim writes."""
    return max(int(lib.rx_segreduce_grid(n, dim)), 1)


def _sq_buffer(n: int, dim: int, device, sq_out: Optional[Tensor]) -> Tensor:
    need = segreduce_grid(n, dim)
    if sq_out is None:
        return torch.empty(need, dtype=torch.float32, device=device)
    _req(sq_out, torch.float32, "sq_out")
    if sq_out.numel() != need or not sq_out.is_contiguous():
        raise ValueError(f"sq_out: expected {need} contiguous floats")
    return sq_out


def segreduce(occ: Tensor, si: SortedIndex, mode: int, target: Tensor, *, state: Optional[Tensor] = None,
              lr: float = 0.0, eps: float = 0.0, scale: Optional[Tensor] = None,
              want_sq: bool = False, occ_ld: Optional[int] = None, dim: Optional[int] = None,
              occ2: Optional[Tensor] = None, target2: Optional[Tensor] = None,
              state2: Optional[Tensor] = None, sq_out: Optional[Tensor] = None) -> Optional[Tensor]:
    """Run rx_segreduce.  ``occ`` is [n, dim] (or 1-D for dim == 1); ``target`` the table / dense
    gradient / compact buffer.  ``occ2`` / ``target2`` / ``state2``: companion scalar table on the
    same rows.  Returns the per-block sum-of-squares partials when asked (written to ``sq_out``, a
    buffer of ``segreduce_grid(n, dim)`` floats, when that is given)."""
    if occ2 is not None:
        _req(occ2, torch.float32, "companion gradients"); _req(target2, torch.float32, "companion target")
        if occ2.numel() < si.n:
            raise ValueError("companion gradients: too few elements")
    _req(occ, torch.float32, "occurrence gradients")
    _req(target, torch.float32, "target")
    if dim is None:
        dim = occ.shape[1] if occ.dim() == 2 else 1
    if occ_ld is None:
        occ_ld = dim
    if occ.numel() < (si.n - 1) * occ_ld + dim and si.n > 0:
        raise ValueError("occurrence gradients: too few elements for the index list")
    sq = _sq_buffer(si.n, dim, occ.device, sq_out) if want_sq else None
    ws = _ws(int(lib.rx_segreduce_workspace_bytes(si.n, dim)), occ.device)
    check(lib.rx_segreduce(_p(occ), occ_ld, si.n, dim, _p(si.sorted_pos), _p(si.sorted_slot), _p(si.uniq_rows),
                           _p(si.seg_off), _p(si.n_uniq), mode, _p(target), _p(state), _p(occ2), _p(target2),
                           _p(state2), si.n_rows, lr, eps, _p(scale), _p(sq), _p(ws), ws.numel(), _dev(occ),
                           _stream(occ)), "rx_segreduce")
    return sq


def segreduce_fm(tsum: Tensor, g: Tensor, n_fields_ex: int, si: SortedIndex, mode: int, target: Tensor, *,
                 table: Optional[Tensor] = None, state: Optional[Tensor] = None, target2: Optional[Tensor] = None,
                 state2: Optional[Tensor] = None, lr: float = 0.0, eps: float = 0.0,
                 scale: Optional[Tensor] = None, want_sq: bool = False,
                 sq_out: Optional[Tensor] = None, deep: Optional[Tensor] = None) -> Optional[Tensor]:
    """rx_segreduce_fm: N == 1 FM backward straight from T [B, D] and g [B] (no per-occurrence rows).
    ``deep`` [B, >= (1+F)*D]: gradient of the tower input laid out by fm2_fwd; its per-field slices
    are added to the rows' gradients in the same walk."""
    _req(tsum, torch.float32, "T", 2); _req(g, torch.float32, "g"); _req(target, torch.float32, "target")
    for t, name in ((table, "table"), (target2, "companion target")):
        if t is not None:
            _req(t, torch.float32, name)     # rows back to back: spaced-out tables take the field-grouped path only
    B, dim = tsum.shape
    deep_ld = 0
    if deep is not None:
        deep, deep_ld = _mat(deep, "deep")
        if deep.shape[0] != B or deep.shape[1] < (n_fields_ex + 1) * dim:
            raise ValueError("deep: expected [B, >= (1+F)*dim]")
    if g.numel() != B or si.n != B * (n_fields_ex + 1):
        raise ValueError("segreduce_fm: shapes disagree with the sorted occurrence list")
    sq = _sq_buffer(si.n, dim, tsum.device, sq_out) if want_sq else None
    ws = _ws(int(lib.rx_segreduce_workspace_bytes(si.n, dim)), tsum.device)
    check(lib.rx_segreduce_fm(_p(tsum), tsum.stride(0), _p(g), _p(deep), deep_ld, B, n_fields_ex, dim, _p(si.sorted_pos),
                              _p(si.sorted_slot), _p(si.uniq_rows), _p(si.seg_off), _p(si.n_uniq), mode, _p(target),
                              _p(state), _p(table), _p(target2), _p(state2), si.n_rows, lr, eps, _p(scale), _p(sq),
                              _p(ws), ws.numel(), _dev(tsum), _stream(tsum)), "rx_segreduce_fm")
    return sq


# ------------------------------------------------------------------ NeuMF train step in one launch
def neumf_step_supported(dim: int, n_items_per_row: int) -> bool:
    return bool(lib.rx_neumf_step_supported(int(dim), int(n_items_per_row)))


def neumf_step(tables, users: Tensor, items: Tensor, labels: Tensor, weights, p_drop: float = 0.0,
               seed: Optional[Tensor] = None, want_logits: bool = False):
    """rx_neumf_step: forward, BCE loss and backward of the list-wise NeuMF (embedding_dim 64) in one launch.
    ``tables`` = (user_gmf, item_gmf, user_mlp, item_mlp) weights, ``weights`` = (w1, b1, w2, b2, wg, bg, wf, bf)
    of linear_mlp.0, linear_mlp.3, linear_gmf, linear_final.  Returns (loss fp64 scalar, du_gmf [B, 64],
    du_mlp [B, 64], di_gmf [B * N, 64], di_mlp [B * N, 64], flat weight gradients, logits [B, N] or None)."""
    ug, ig, um, im = (_table(t, "embedding table") for t in tables)
    w1, b1, w2, b2, wg, bg, wf, bf = (_req(t, torch.float32, "tower weight") for t in weights)
    D = ug.shape[1]
    if not (ig.shape[1] == D and um.shape == ug.shape and im.shape == ig.shape):
        raise ValueError("neumf_step: the GMF and MLP tables must have equal shapes")
    if tuple(w1.shape) != (D, 2 * D) or tuple(w2.shape) != (D // 2, D) or tuple(wg.shape) != (D // 2, D) or wf.numel() != D \
            or b1.numel() != D or b2.numel() != D // 2 or bg.numel() != D // 2 or bf.numel() != 1:
        raise ValueError("neumf_step: tower shapes must be 2D -> D -> D/2, GMF D -> D/2, final D -> 1")
    users, items = _idx(users, "user_code"), _idx(items, "item_code")
    if users.dim() != 1 or items.dim() != 2 or items.shape[0] != users.shape[0]:
        raise ValueError("neumf_step: user_code [B] and item_code [B, N]")
    B, N = items.shape
    if labels.dtype not in (torch.float64, torch.float32):
        raise TypeError(f"label: expected float64 or float32, got {labels.dtype}")
    _req(labels, labels.dtype, "label")
    if labels.numel() != B * N:
        raise ValueError(f"label: expected {B * N} elements")
    _same_device(ug, ig, um, im, users, items, labels, w1, b1, w2, b2, wg, bg, wf, bf, seed)
    dev = ug.device
    _err_flag(dev)          # (first use of the device: library set-up)
    if seed is not None:
        _req(seed, torch.int64, "seed")
    du_g = torch.empty(B, D, dtype=torch.float32, device=dev)
    du_m = torch.empty(B, D, dtype=torch.float32, device=dev)
    di_g = torch.empty(B * N, D, dtype=torch.float32, device=dev)
    di_m = torch.empty(B * N, D, dtype=torch.float32, device=dev)
    wgrad = torch.empty(int(lib.rx_neumf_step_weight_floats()), dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float64, device=dev)
    logits = torch.empty(B, N, dtype=torch.float32, device=dev) if want_logits else None
    a = _lib.NeuMFArgs()
    a.user_gmf, a.item_gmf, a.user_mlp, a.item_mlp = ug.data_ptr(), ig.data_ptr(), um.data_ptr(), im.data_ptr()
    a.n_users, a.n_items = ug.shape[0], ig.shape[0]
    a.users, a.items, a.labels = users.data_ptr(), items.data_ptr(), labels.data_ptr()
    a.labels_f64, a.N, a.B, a.dim = int(labels.dtype == torch.float64), N, B, D
    a.w1, a.b1, a.w2, a.b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
    a.wg, a.bg, a.wf, a.bf = wg.data_ptr(), bg.data_ptr(), wf.data_ptr(), bf.data_ptr()
    a.p_drop = float(p_drop)
    a.seed = seed.data_ptr() if seed is not None else 0
    a.du_gmf, a.du_mlp, a.di_gmf, a.di_mlp = du_g.data_ptr(), du_m.data_ptr(), di_g.data_ptr(), di_m.data_ptr()
    a.wgrad, a.loss, a.logits = wgrad.data_ptr(), loss.data_ptr(), (logits.data_ptr() if logits is not None else 0)
    ws = _ws(int(lib.rx_neumf_step_workspace_bytes(B, N)), dev)
    check(lib.rx_neumf_step(C.byref(a), _p(ws), ws.numel(), _p(_err_flag(dev)), _dev(ug), _stream(ug)), "rx_neumf_step")
    _after_index_op(dev)
    return loss, du_g, du_m, di_g, di_m, wgrad, logits


# ------------------------------------------------------------------ field-grouped backward (N == 1 field models)
@dataclass
class FieldGroups:
    """Result of rx_fg_group: the occurrences of an [B, n_fields] index matrix, field after field,
    each field ordered by (row & 255, row >> 8, sample).  ``srow`` / ``sb`` are int32 tensors holding
    uint32 values (table row and sample of every position)."""
    n: int
    n_rows: int
    B: int
    n_fields: int
    srow: Tensor
    sb: Tensor

    def distinct_rows(self) -> Tensor:
        """The batch's distinct table rows (int64, in list order): the last position of every run."""
        r = self.srow.to(torch.int64) & 0xFFFFFFFF
        end = torch.ones_like(r, dtype=torch.bool)
        end[:-1] = r[:-1] != r[1:]
        return r[end]


def _fg_args(ex_idx: Tensor, item_idx: Tensor, field_base: Sequence[int]):
    ex_idx, F, ex_ld = _idx_matrix(ex_idx, "ex_idx")
    item_idx, N, item_ld = _idx_matrix(item_idx, "item_idx")
    nf = F + 1
    if N != 1 or ex_idx.dim() != 2 or item_idx.shape[0] != ex_idx.shape[0]:
        raise ValueError("field grouping needs ex_idx [B, F] and item_idx [B, 1]")
    if len(field_base) != nf + 1 or nf > _lib.RX_FG_MAX_FIELDS:
        raise ValueError(f"field_base: expected {nf + 1} row offsets (at most {_lib.RX_FG_MAX_FIELDS} fields)")
    a = _lib.FGArgs()
    a.ex_idx, a.item_idx, a.ex_ld, a.item_ld = ex_idx.data_ptr(), item_idx.data_ptr(), ex_ld, item_ld
    a.B, a.n_fields = ex_idx.shape[0], nf
    for k, v in enumerate(field_base):
        a.field_base[k] = int(v)
    return a


def fg_supported(ex_idx: Tensor, item_idx: Tensor, field_base: Sequence[int], dim: int) -> bool:
    """Whether the field-grouped backward takes this batch shape, field layout and row width."""
    if item_idx.dim() != 2 or item_idx.shape[1] != 1 or ex_idx.dim() != 2 or ex_idx.shape[0] == 0:
        return False
    if len(field_base) != ex_idx.shape[1] + 2 or len(field_base) - 1 > _lib.RX_FG_MAX_FIELDS:
        return False
    if not lib.rx_fg_dim_supported(int(dim)):
        return False
    return bool(lib.rx_fg_supported(C.byref(_fg_args(ex_idx, item_idx, field_base))))


def fg_group(ex_idx: Tensor, item_idx: Tensor, field_base: Sequence[int], out: Optional[FieldGroups] = None) -> FieldGroups:
    """rx_fg_group.  ``field_base``: host ints, first table row of every field plus the table height.
    ``out``: a FieldGroups of the same shape whose tensors are overwritten in place."""
    _same_device(ex_idx, item_idx)
    a = _fg_args(ex_idx, item_idx, field_base)
    B, nf, dev = int(a.B), int(a.n_fields), ex_idx.device
    n, n_rows = B * nf, int(field_base[-1])
    if out is not None:
        if out.n != n or out.n_rows != n_rows or out.B != B:
            raise ValueError("fg_group: `out` was made for a different batch shape")
        srow, sb = out.srow, out.sb
    else:
        srow = torch.empty(n, dtype=torch.int32, device=dev)
        sb = torch.empty(n, dtype=torch.int32, device=dev)
    ws = _ws(int(lib.rx_fg_workspace_bytes(B, nf)), dev)
    check(lib.rx_fg_group(C.byref(a), _p(srow), _p(sb), _p(ws), ws.numel(), _p(_err_flag(dev)), _dev(ex_idx),
                          _stream(ex_idx)), "rx_fg_group")
    return out if out is not None else FieldGroups(n, n_rows, B, nf, srow, sb)


def _table_strides(table: Optional[Tensor], table2: Optional[Tensor], dim: int):
    """(row stride of the table, stride of its [rows, 1] / [rows] companion) in floats for the rx_fg_* calls."""
    ld_w, ld_w2 = dim, 1
    if table is not None:
        _, ld_w = _row_strided(table, "table", dim)
        if ld_w % 4 or table.data_ptr() % 16:
            raise ValueError("table: row stride must be a multiple of 4 floats, 16-byte aligned")
    if table2 is not None:
        if not (table2.is_cuda and table2.dtype == torch.float32) or table2.numel() != (table2.shape[0] if table2.dim() else 1):
            raise TypeError("companion table: expected fp32 [rows] or [rows, 1] on the GPU")
        ld_w2 = table2.stride(0) if table2.shape[0] > 1 else 1
    return ld_w, ld_w2


def fg_reduce_grid(n: int, dim: int) -> int:
    return max(int(lib.rx_fg_reduce_grid(n, dim)), 1)


def fg_reduce(tsum: Tensor, g: Tensor, fg: FieldGroups, mode: int, target: Tensor, *, table: Optional[Tensor] = None,
              state: Optional[Tensor] = None, target2: Optional[Tensor] = None, state2: Optional[Tensor] = None,
              lr: float = 0.0, eps: float = 0.0, scale: Optional[Tensor] = None, want_sq: bool = False,
              sq_out: Optional[Tensor] = None, deep: Optional[Tensor] = None, premultiplied: bool = False,
              undo: Optional[Tensor] = None, undo2: Optional[Tensor] = None, loss_final=None) -> Optional[Tensor]:
    """rx_fg_reduce: the N == 1 FM backward over a field-grouped occurrence list (see FieldGroups).
    In RX_SEG_COMPACT mode ``target`` / ``target2`` have one row per POSITION (fg.n rows).
    RX_SEG_SGD_SPEC (the clipped SGD step taken at once with factor 1): ``undo`` [fg.n, 2 * dim] and, unless
    the companion entry sits in the row's line, ``undo2`` [fg.n, 2] receive what fg_row_update needs to redo
    the rows when the clip bites.
    ``tsum`` [B, dim] and ``g`` [B] may be column blocks of one wider matrix (unit inner stride for
    ``tsum``, any stride for ``g``); ``premultiplied``: ``tsum`` already holds g_b * T_b."""
    for t, name in ((tsum, "T"), (g, "g")):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
            raise TypeError(f"{name}: expected an fp32 tensor on the GPU")
    if mode in (_lib.RX_SEG_COMPACT, _lib.RX_SEG_DENSE):
        _req(target, torch.float32, "target")     # gradient arrays: rows back to back
    if tsum.dim() != 2 or tsum.stride(1) != 1 or g.dim() != 1:
        raise ValueError("fg_reduce: T must be [B, dim] with unit inner stride, g a vector")
    B, dim = tsum.shape
    deep_ld = 0
    if deep is not None:
        deep, deep_ld = _mat(deep, "deep")
        if deep.shape[0] != B or deep.shape[1] < fg.n_fields * dim:
            raise ValueError("deep: expected [B, >= n_fields * dim]")
    if g.numel() != B or fg.B != B:
        raise ValueError("fg_reduce: shapes disagree with the grouped occurrence list")
    if mode == _lib.RX_SEG_COMPACT and target.shape[0] < fg.n:
        raise ValueError("fg_reduce: the compact gradient needs one row per position")
    sq = None
    if want_sq:
        need = fg_reduce_grid(fg.n, dim)
        if sq_out is None:
            sq = torch.empty(need, dtype=torch.float32, device=tsum.device)
        else:
            _req(sq_out, torch.float32, "sq_out")
            if sq_out.numel() != need:
                raise ValueError(f"sq_out: expected {need} contiguous floats")
            sq = sq_out
    a = _lib.FGReduceArgs()
    a.T, a.ld_T = tsum.data_ptr(), tsum.stride(0) if B > 1 else max(dim, tsum.stride(0))
    a.g, a.ld_g = g.data_ptr(), (g.stride(0) if B > 1 else 1)
    a.deep, a.deep_ld = (deep.data_ptr() if deep is not None else 0), deep_ld
    a.B, a.n_fields, a.dim, a.mode, a.premultiplied = B, fg.n_fields, dim, mode, int(premultiplied)
    a.srow, a.sb = fg.srow.data_ptr(), fg.sb.data_ptr()
    for name, t in (("target", target), ("state", state), ("table", table), ("target2", target2), ("state2", state2),
                    ("scale", scale), ("sq_part", sq)):
        setattr(a, name, 0 if t is None else t.data_ptr())
    a.n_rows, a.lr, a.eps = fg.n_rows, lr, eps
    if mode == _lib.RX_SEG_SGD_SPEC:
        _req(undo, torch.float32, "undo", 2)
        if undo.shape[0] < fg.n or undo.shape[1] != 2 * dim:
            raise ValueError("fg_reduce: undo must be [positions, 2 * dim]")
        if undo2 is not None:
            _req(undo2, torch.float32, "undo2", 2)
            if undo2.shape[0] < fg.n or undo2.shape[1] != 2:
                raise ValueError("fg_reduce: undo2 must be [positions, 2]")
        a.undo, a.undo2 = undo.data_ptr(), (undo2.data_ptr() if undo2 is not None else 0)
    global _loss_final_pending
    if loss_final is None and _loss_final_pending is not None:
        loss_final, _loss_final_pending = _loss_final_pending, None
    if loss_final is not None:      # (partials, their count, number of logits, loss tensor) of fm2_fwd_bce(defer_final=True)
        parts, n_parts, count, loss_out = loss_final
        a.loss_parts, a.n_loss_parts, a.loss_count, a.loss_out = parts.data_ptr(), n_parts, count, loss_out.data_ptr()
    grad_mode = mode in (_lib.RX_SEG_COMPACT, _lib.RX_SEG_DENSE)
    a.ld_w, a.ld_w2 = _table_strides(table if grad_mode else target, None if grad_mode else target2, dim)
    ws = _ws(int(lib.rx_fg_reduce_workspace_bytes(fg.n, dim)), tsum.device)
    check(lib.rx_fg_reduce(C.byref(a), _p(ws), ws.numel(), _dev(tsum), _stream(tsum)), "rx_fg_reduce")
    return sq


# The final sum of a forward pass's loss partials, handed to the next fg_reduce call (see fm2_fwd_bce(defer_final=True)).
_loss_final_pending = None


def fm2_fwd_bce(feat, bias, ex_idx, ex_base, item_idx, item_base: int, labels: Tensor, want_logits: bool = False,
                defer_final: bool = False):
    """rx_fm2_fwd_bce: logits, BCE loss and the per-sample backward record from one forward kernel (N == 1).
    Returns (loss fp64 scalar, gt [B, ld] fp32 with gt[:, :dim] = g_b * T_b and gt[:, dim] = g_b,
    logits [B, 1] or None).  ``defer_final``: the one-wavefront launch that adds the block partials is left
    out; the returned loss is written by the ``fg_reduce`` call of the backward pass instead, to which the
    caller hands the fourth return value (``fg_reduce(..., loss_final=)``)."""
    a, B, F, N = _fm_args(feat, bias, ex_idx, ex_base, item_idx, item_base)
    if N != 1:
        raise ValueError("fm2_fwd_bce: one item per sample only")
    if labels.dtype not in (torch.float64, torch.float32):
        raise TypeError(f"label: expected float64 or float32, got {labels.dtype}")
    _req(labels, labels.dtype, "label")
    if labels.numel() != B:
        raise ValueError(f"label: expected {B} elements")
    _same_device(feat, labels)
    dim, dev = feat.shape[1], feat.device
    ld = (dim + 1 + 31) // 32 * 32          # whole 128-byte lines per sample
    gt = torch.empty(B, ld, dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float64, device=dev)
    logits = torch.empty(B, 1, dtype=torch.float32, device=dev) if want_logits else None
    ws = _ws(int(lib.rx_fm2_fwd_bce_workspace_bytes(B, dim)), dev)
    check(lib.rx_fm2_fwd_bce(C.byref(a), _p(labels), int(labels.dtype == torch.float64), _p(logits), _p(gt), ld,
                             _p(None if defer_final else loss), _p(ws), ws.numel(), _p(_err_flag(dev)), _dev(feat), _stream(feat)),
          "rx_fm2_fwd_bce")
    _after_index_op(dev)
    if defer_final:
        return loss, gt, logits, (ws, int(lib.rx_fm2_fwd_bce_partials(B, dim)), B, loss)
    return loss, gt, logits


def fg_row_update(compact: Tensor, fg: FieldGroups, mode: int, table: Tensor, *, state=None, lr=0.0, eps=0.0,
                  scale=None, compact2=None, table2=None, state2=None, clip_parts=None, max_norm=None):
    """``clip_parts`` / ``max_norm``: instead of a ready ``scale``, ALL of the step's partial sums of
    squares; the launch forms the clip factor itself (rx_fg_row_update_clip) and (scale, norm) are
    returned as one-element tensors."""
    _req(compact, torch.float32, "compact", 2)
    dim = compact.shape[1] // 2 if mode in (_lib.RX_SEG_SGD_SPEC, _lib.RX_SEG_SGD_SPEC_REDO) else compact.shape[1]    # the speculative step's undo records: gradient | old row
    ld_w, ld_w2 = _table_strides(table, table2, dim)
    if clip_parts is not None:
        _req(clip_parts, torch.float32, "clip_parts")
        if scale is not None or not clip_parts.is_contiguous():
            raise ValueError("fg_row_update: give either a scale or contiguous partial sums")
        out = torch.empty(2, dtype=torch.float32, device=table.device)
        check(lib.rx_fg_row_update_clip(_p(compact), dim, _p(fg.srow), fg.n, mode, _p(table), _p(state),
                                        _p(compact2), _p(table2), _p(state2), fg.n_rows, lr, eps, _p(clip_parts),
                                        clip_parts.numel(), float(max_norm), _p(out[0:1]), _p(out[1:2]), ld_w, ld_w2,
                                        _dev(table), _stream(table)), "rx_fg_row_update_clip")
        return out[0:1], out[1:2]
    check(lib.rx_fg_row_update(_p(compact), dim, _p(fg.srow), fg.n, mode, _p(table), _p(state),
                               _p(compact2), _p(table2), _p(state2), fg.n_rows, lr, eps, _p(scale), ld_w, ld_w2,
                               _dev(table), _stream(table)), "rx_fg_row_update")


def row_update(compact: Tensor, si: SortedIndex, mode: int, table: Tensor, *, state=None, lr=0.0, eps=0.0,
               scale=None, compact2=None, table2=None, state2=None) -> None:
    """``compact2`` [n] / ``table2`` / ``state2``: the companion scalar table, updated in the same launch.
    Modes RX_SEG_SGD_SPEC / _REDO: ``compact`` [n, 2 * dim] and ``compact2`` [n, 2] are the undo records the
    reduce left (segreduce / segreduce_fm with ``state=`` / ``state2=`` in that mode)."""
    dim = compact.shape[1] if compact.dim() == 2 else 1
    if mode in (_lib.RX_SEG_SGD_SPEC, _lib.RX_SEG_SGD_SPEC_REDO):
        if compact.dim() != 2 or compact.shape[1] % 2 or scale is None:
            raise ValueError("row_update: the speculative step takes [n, 2 * dim] undo records and a clip factor")
        dim = compact.shape[1] // 2
    _req(table, torch.float32, "table")
    if table2 is not None:
        _req(table2, torch.float32, "companion table")
    check(lib.rx_row_update(_p(compact), dim, _p(si.uniq_rows), _p(si.n_uniq), si.n, mode, _p(table), _p(state),
                            _p(compact2), _p(table2), _p(state2), si.n_rows, lr, eps, _p(scale), _dev(table),
                            _stream(table)), "rx_row_update")


@dataclass
class TableReduce:
    """One table's share of a multi-table reduce / update (rx_seg_table in recsys_hip.h).  For
    ``segreduce_multi``: ``occ`` the per-occurrence gradients, ``target`` the table / dense gradient /
    compact buffer.  For ``row_update_multi``: ``occ`` the compact gradients, ``target`` the table."""
    occ: Tensor
    si: SortedIndex
    target: Tensor
    dim: int
    occ_ld: Optional[int] = None
    state: Optional[Tensor] = None
    occ2: Optional[Tensor] = None
    target2: Optional[Tensor] = None
    state2: Optional[Tensor] = None
    sq: Optional[Tensor] = None


def _seg_tables(items: Sequence[TableReduce], compact: bool):
    from ._lib import SegTable, RX_SEG_MULTI_MAX
    if not 1 <= len(items) <= RX_SEG_MULTI_MAX:
        raise ValueError(f"1..{RX_SEG_MULTI_MAX} tables per launch")
    arr = (SegTable * len(items))()
    for k, t in enumerate(items):
        _req(t.occ, torch.float32, "gradients"); _req(t.target, torch.float32, "target")
        ld = t.dim if (compact or t.occ_ld is None) else t.occ_ld
        if t.si.n > 0 and t.occ.numel() < (t.si.n - 1) * ld + t.dim and not compact:
            raise ValueError("occurrence gradients: too few elements for the index list")
        if t.occ2 is not None:
            _req(t.occ2, torch.float32, "companion gradients"); _req(t.target2, torch.float32, "companion target")
        if t.sq is not None and (t.sq.numel() != segreduce_grid(t.si.n, t.dim) or not t.sq.is_contiguous()):
            raise ValueError("sq: expected segreduce_grid(n, dim) contiguous floats")
        arr[k] = SegTable(_p(t.occ), ld, t.si.n, t.dim, _p(t.si.sorted_pos), _p(t.si.sorted_slot), _p(t.si.uniq_rows),
                          _p(t.si.seg_off), _p(t.si.n_uniq), _p(t.target), _p(t.state), _p(t.occ2), _p(t.target2),
                          _p(t.state2), t.si.n_rows, _p(t.sq))
    return arr


def segreduce_multi(items: Sequence[TableReduce], mode: int, *, lr: float = 0.0, eps: float = 0.0,
                    scale: Optional[Tensor] = None) -> None:
    """rx_segreduce for several tables of one lane layout (equal widths always qualify) in one pair
    of launches; results equal one ``segreduce`` per table bit for bit."""
    arr = _seg_tables(items, compact=False)
    dev = items[0].occ.device
    ws = _ws(sum(int(lib.rx_segreduce_workspace_bytes(t.si.n, t.dim)) for t in items), dev)
    check(lib.rx_segreduce_multi(arr, len(items), mode, lr, eps, _p(scale), _p(ws), ws.numel(), _dev(items[0].occ),
                                 _stream(items[0].occ)), "rx_segreduce_multi")


def row_update_multi(items: Sequence[TableReduce], mode: int, *, lr: float = 0.0, eps: float = 0.0,
                     scale: Optional[Tensor] = None) -> None:
    """rx_row_update for several tables in one launch (``occ`` / ``occ2``: compact gradients)."""
    arr = _seg_tables(items, compact=True)
    check(lib.rx_row_update_multi(arr, len(items), mode, lr, eps, _p(scale), _dev(items[0].target),
                                  _stream(items[0].target)), "rx_row_update_multi")


def step_tail(items: Sequence[TableReduce], mode: int, lr: float, eps: float, dense_ws, dense_gs, dense_states, dense_mode: int,
              dense_lr: float, dense_eps: float, parts: Optional[Tensor], parts64: Optional[Tensor], max_norm: float):
    """rx_step_tail: clip factor + row updates of several tables + update of the dense tensors, one launch.
    ``items`` as for ``row_update_multi``; returns (scale, total norm) as one-element tensors."""
    arr = _seg_tables(items, compact=True)
    dev = items[0].target.device
    t = _tensor_list(dense_ws, dense_gs, dense_states) if dense_gs else None
    if parts is not None:
        _req(parts, torch.float32, "parts")
        if not parts.is_contiguous():
            raise ValueError("parts: contiguous partial sums expected")
    if parts64 is not None:
        _req(parts64, torch.float64, "parts64")
    scale = torch.empty(1, dtype=torch.float32, device=dev)
    total = torch.empty(1, dtype=torch.float32, device=dev)
    check(lib.rx_step_tail(arr, len(items), mode, lr, eps, C.byref(t) if t is not None else None, dense_mode, dense_lr, dense_eps,
                           _p(parts), parts.numel() if parts is not None else 0, _p(parts64),
                           parts64.numel() if parts64 is not None else 0, max_norm, _p(scale), _p(total), _dev(items[0].target),
                           _stream(items[0].target)), "rx_step_tail")
    return scale, total


# ------------------------------------------------------------------ dense optimizers / reductions
def dense_adam(w, g, m, v, lr, beta1, beta2, eps, step, scale=None):
    """``step``: the step count t >= 1 as an int, or an int32 device tensor holding it (the form a
    recorded graph needs: the counter is advanced on the stream, the recording reads it)."""
    if isinstance(step, Tensor):
        _req(step, torch.int32, "step")
        check(lib.rx_dense_adam_dstep(_p(w), _p(g), _p(m), _p(v), w.numel(), lr, beta1, beta2, eps, _p(step),
                                      _p(scale), _dev(w), _stream(w)), "rx_dense_adam_dstep")
        return
    check(lib.rx_dense_adam(_p(w), _p(g), _p(m), _p(v), w.numel(), lr, beta1, beta2, eps, step, _p(scale),
                            _dev(w), _stream(w)), "rx_dense_adam")


def dense_sgd(w, g, lr, scale=None):
    check(lib.rx_dense_sgd(_p(w), _p(g), w.numel(), lr, _p(scale), _dev(w), _stream(w)), "rx_dense_sgd")


def dense_adagrad(w, g, state, lr, eps, scale=None):
    check(lib.rx_dense_adagrad(_p(w), _p(g), _p(state), w.numel(), lr, eps, _p(scale), _dev(w), _stream(w)),
          "rx_dense_adagrad")


def sumsq_accumulate(x: Tensor, acc: Tensor) -> None:
    _req(x, torch.float32, "x"); _req(acc, torch.float64, "acc")
    ws = _ws(int(lib.rx_sumsq_workspace_bytes(x.numel())), x.device)
    check(lib.rx_sumsq_accumulate(_p(x), x.numel(), _p(acc), _p(ws), ws.numel(), _dev(x), _stream(x)),
          "rx_sumsq_accumulate")


def sum_accumulate(x: Tensor, acc: Tensor) -> None:
    _req(x, torch.float32, "x"); _req(acc, torch.float64, "acc")
    ws = _ws(int(lib.rx_sumsq_workspace_bytes(x.numel())), x.device)
    check(lib.rx_sum_accumulate(_p(x), x.numel(), _p(acc), _p(ws), ws.numel(), _dev(x), _stream(x)),
          "rx_sum_accumulate")


def clip_scale(acc: Tensor, max_norm: float):
    scale = torch.empty(1, dtype=torch.float32, device=acc.device)
    total = torch.empty(1, dtype=torch.float32, device=acc.device)
    check(lib.rx_clip_scale(_p(acc), max_norm, _p(scale), _p(total), _dev(acc), _stream(acc)), "rx_clip_scale")
    return scale, total


def clip_scale_parts(parts: Optional[Tensor], acc: Optional[Tensor], max_norm: float, parts64: Optional[Tensor] = None):
    """Clip factor from per-block partial sums of squares: fp32 ``parts`` (segmented reduce),
    fp64 ``parts64`` (multi_sumsq_partials), plus ``acc``, a device double, if given."""
    ref = parts if parts is not None else parts64
    if parts is not None:
        _req(parts, torch.float32, "parts")
    if parts64 is not None:
        _req(parts64, torch.float64, "parts64")
    scale = torch.empty(1, dtype=torch.float32, device=ref.device)
    total = torch.empty(1, dtype=torch.float32, device=ref.device)
    check(lib.rx_clip_scale_parts(_p(parts), parts.numel() if parts is not None else 0, _p(parts64),
                                  parts64.numel() if parts64 is not None else 0, _p(acc), max_norm, _p(scale),
                                  _p(total), _dev(ref), _stream(ref)), "rx_clip_scale_parts")
    return scale, total


def _tensor_list(ws, gs, states):
    from ._lib import TensorList, RX_MULTI_MAX
    if len(gs) > RX_MULTI_MAX:
        raise ValueError(f"at most {RX_MULTI_MAX} tensors per launch")
    t = TensorList()
    t.count = len(gs)
    for k, g in enumerate(gs):
        _req(g, torch.float32, "g")
        t.g[k] = g.data_ptr()
        t.n[k] = g.numel()
        if ws is not None:
            w = ws[k]
            if not (w.is_contiguous() and w.numel() == g.numel() and w.dtype == torch.float32):
                raise ValueError("weight / gradient mismatch")
            t.w[k] = w.data_ptr()
        if states is not None:
            t.state[k] = states[k].data_ptr()
    return t


def multi_sumsq_partials(gs) -> Tensor:
    """fp64 partial sums of squares of up to 24 contiguous fp32 tensors, one launch."""
    t = _tensor_list(None, gs, None)
    out = torch.empty(int(lib.rx_multi_sumsq_partials(len(gs))), dtype=torch.float64, device=gs[0].device)
    check(lib.rx_multi_sumsq(C.byref(t), _p(out), _dev(gs[0]), _stream(gs[0])), "rx_multi_sumsq")
    return out


def multi_dense_update(ws, gs, mode: int, lr: float, eps: float = 0.0, scale=None, states=None) -> None:
    """SGD / Adagrad step of up to 24 small dense tensors in one launch."""
    t = _tensor_list(ws, gs, states)
    check(lib.rx_multi_dense_update(C.byref(t), mode, lr, eps, _p(scale), _dev(ws[0]), _stream(ws[0])),
          "rx_multi_dense_update")


def multi_dense_adam(ws, gs, ms, vs, lr: float, beta1: float, beta2: float, eps: float, step, scale=None) -> None:
    """Adam step of up to 24 dense tensors in one launch (rx_multi_dense_adam); ``step`` as for ``dense_adam``."""
    if not 1 <= len(ws) <= _lib.RX_MULTI_MAX or not (len(gs) == len(ms) == len(vs) == len(ws)):
        raise ValueError(f"1..{_lib.RX_MULTI_MAX} tensors with a gradient and two moments each")
    t = _lib.AdamList()
    for k, (w, g, m, v) in enumerate(zip(ws, gs, ms, vs)):
        for x, name in ((w, "parameter"), (g, "gradient"), (m, "first moment"), (v, "second moment")):
            _req(x, torch.float32, name)
            if x.numel() != w.numel():
                raise ValueError(f"{name}: expected {w.numel()} elements")
        _same_device(w, g, m, v)
        t.w[k], t.g[k], t.m[k], t.v[k], t.n[k] = w.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), w.numel()
    t.count = len(ws)
    d_step = None
    if isinstance(step, Tensor):
        _req(step, torch.int32, "step")
        d_step, step = step, 0
    check(lib.rx_multi_dense_adam(C.byref(t), lr, beta1, beta2, eps, int(step), _p(d_step), _p(scale), _dev(ws[0]), _stream(ws[0])),
          "rx_multi_dense_adam")


def reduce_sum(x: Tensor) -> Tensor:
    _req(x, torch.float32, "x")
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = _ws(int(lib.rx_sumsq_workspace_bytes(x.numel())), x.device)
    check(lib.rx_reduce_sum(_p(x), x.numel(), _p(out), _p(ws), ws.numel(), _dev(x), _stream(x)), "rx_reduce_sum")
    return out


# ------------------------------------------------------------------ dense layers
def _mat(t: Tensor, name: str) -> Tuple[Tensor, int]:
    """2-D fp32 with unit inner stride (a column block of a wider matrix is allowed)."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2):
        raise TypeError(f"{name}: expected a 2-D fp32 GPU tensor")
    if t.stride(1) != 1 and t.shape[1] != 1:
        raise ValueError(f"{name}: inner stride must be 1")
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    return t, ld


def linear_fwd(x: Tensor, w: Tensor, b: Optional[Tensor], act: int = 0, bf16: bool = False,
               out: Optional[Tensor] = None) -> Tensor:
    x, ldx = _mat(x, "x")
    _req(w, torch.float32, "weight", 2)
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"linear: x is [{M},{K}] but weight is {tuple(w.shape)}")
    if b is not None:
        _req(b, torch.float32, "bias", 1)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    out, ldy = _mat(out, "out")
    check(lib.rx_linear_fwd(_p(x), ldx, _p(w), _p(b), _p(out), ldy, M, N, K, act, int(bf16), _dev(x), _stream(x)),
          "rx_linear_fwd")
    return out


def linear_bwd_x(dy: Tensor, w: Tensor, bf16: bool = False, out: Optional[Tensor] = None,
                 accumulate: bool = False) -> Tensor:
    dy, ldd = _mat(dy, "dy")
    M, N = dy.shape
    K = w.shape[1]
    if out is None:
        # rows padded to a multiple of 4 floats so that consumers can use 16-byte accesses
        kp = (K + 3) // 4 * 4
        buf = torch.empty(M, kp, dtype=torch.float32, device=dy.device)
        out = buf if kp == K else buf[:, :K]
    out, ldx = _mat(out, "dx")
    check(lib.rx_linear_bwd_x(_p(dy), ldd, _p(w), _p(out), ldx, M, N, K, int(accumulate), int(bf16), _dev(dy),
                              _stream(dy)), "rx_linear_bwd_x")
    return out


def linear_bwd_w(dy: Tensor, x: Tensor, want_bias: bool = True, bf16: bool = False):
    dy, ldd = _mat(dy, "dy")
    x, ldx = _mat(x, "x")
    M, N = dy.shape
    K = x.shape[1]
    dw = torch.empty(N, K, dtype=torch.float32, device=dy.device)
    db = torch.empty(N, dtype=torch.float32, device=dy.device) if want_bias else None
    ws = _ws(int(lib.rx_linear_bwd_w_workspace_bytes(M, N, K)), dy.device)
    check(lib.rx_linear_bwd_w(_p(dy), ldd, _p(x), ldx, _p(dw), _p(db), M, N, K, int(bf16), _p(ws), ws.numel(),
                              _dev(dy), _stream(dy)), "rx_linear_bwd_w")
    return dw, db


def relu_bwd(dy: Tensor, y: Tensor, scale: float = 1.0) -> Tensor:
    """dz = y > 0 ? dy * scale : 0.  ``scale`` != 1: backward of ReLU + Dropout from the pair's output
    ``y`` (positive exactly where the ReLU was open and the element kept), scale = 1 / (1 - p)."""
    _req(dy, torch.float32, "dy"); _req(y, torch.float32, "y")
    if dy.numel() != y.numel():
        raise ValueError("relu_bwd: dy and y differ in size")
    dz = torch.empty_like(dy)
    if scale != 1.0:
        check(lib.rx_relu_bwd_scaled(_p(dy), _p(y), _p(dz), dy.numel(), float(scale), _dev(dy), _stream(dy)), "rx_relu_bwd_scaled")
    else:
        check(lib.rx_relu_bwd(_p(dy), _p(y), _p(dz), dy.numel(), _dev(dy), _stream(dy)), "rx_relu_bwd")
    return dz


# ------------------------------------------------------------------ BatchNorm1d (+ ReLU + dropout)
def bn_stats(x: Tensor, eps: float, momentum: float = 0.0, running_mean: Optional[Tensor] = None,
             running_var: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Per-column mean and 1/sqrt(biased var + eps) of x [M, C]; updates the running buffers in place."""
    x, ldx = _mat(x, "x")
    M, C = x.shape
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = _ws(int(lib.rx_bn_workspace_bytes(M, C)), x.device)
    check(lib.rx_bn_stats(_p(x), ldx, M, C, eps, momentum, _p(mean), _p(rstd), _p(running_mean), _p(running_var),
                          _p(ws), ws.numel(), _dev(x), _stream(x)), "rx_bn_stats")
    return mean, rstd


def bn_apply(x: Tensor, mean: Tensor, rstd: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], relu: bool,
             keep: Optional[Tensor] = None, keep_scale: float = 1.0) -> Tensor:
    x, ldx = _mat(x, "x")
    M, C = x.shape
    if keep is not None and not (keep.dtype == torch.uint8 and keep.is_contiguous() and tuple(keep.shape) == (M, C)):
        raise ValueError("keep: expected a contiguous uint8 [M, C] mask")
    out = _padded(M, C, x.device)
    check(lib.rx_bn_apply(_p(x), ldx, M, C, _p(mean), _p(rstd), _p(gamma), _p(beta), int(relu), _p(keep), keep_scale,
                          _p(out), out.stride(0) if M > 1 else max(C, out.stride(0)), _dev(x), _stream(x)), "rx_bn_apply")
    return out


def bn_bwd(dout: Tensor, x: Tensor, out: Optional[Tensor], mean: Tensor, rstd: Tensor, gamma: Optional[Tensor],
           gate: bool, keep_scale: float, batch_stats: bool):
    """Returns (dx, dgamma, dbeta)."""
    dout, ldd = _mat(dout, "dout")
    x, ldx = _mat(x, "x")
    M, C = x.shape
    ldo = 0
    if gate:
        out, ldo = _mat(out, "out")
    dx = _padded(M, C, x.device)
    dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = _ws(int(lib.rx_bn_workspace_bytes(M, C)), x.device)
    check(lib.rx_bn_bwd(_p(dout), ldd, _p(x), ldx, _p(out) if gate else None, ldo, M, C, _p(mean), _p(rstd), _p(gamma),
                        int(gate), keep_scale, int(batch_stats), _p(dx), dx.stride(0) if M > 1 else max(C, dx.stride(0)),
                        _p(dgamma), _p(dbeta), _p(ws), ws.numel(), _dev(x), _stream(x)), "rx_bn_bwd")
    return dx, dgamma, dbeta


def _padded(rows: int, cols: int, device) -> Tensor:
    """fp32 [rows, cols] whose row stride is a multiple of 4 floats."""
    cp = (cols + 3) // 4 * 4
    buf = torch.empty(rows, cp, dtype=torch.float32, device=device)
    return buf if cp == cols else buf[:, :cols]


def cross_combine_fwd(x0, u, xl):
    x0, ld0 = _mat(x0, "x0"); u, ldu = _mat(u, "u"); xl, ldl = _mat(xl, "xl")
    rows, cols = x0.shape
    xn = _padded(rows, cols, x0.device)
    xn, ldn = _mat(xn, "xn")
    check(lib.rx_cross_combine_fwd(_p(x0), ld0, _p(u), ldu, _p(xl), ldl, _p(xn), ldn, rows, cols, _dev(x0),
                                   _stream(x0)), "rx_cross_combine_fwd")
    return xn


def cross_combine_bwd(dxn, x0, u, dx0_acc):
    dxn, ldd = _mat(dxn, "dxn"); x0, ld0 = _mat(x0, "x0"); u, ldu = _mat(u, "u")
    dx0_acc, ldx0 = _mat(dx0_acc, "dx0")
    rows, cols = dxn.shape
    du = _padded(rows, cols, dxn.device)
    dxl = _padded(rows, cols, dxn.device)
    _, lddu = _mat(du, "du")
    _, ldxl = _mat(dxl, "dxl")
    check(lib.rx_cross_combine_bwd(_p(dxn), ldd, _p(x0), ld0, _p(u), ldu, _p(du), lddu, _p(dx0_acc), ldx0,
                                   _p(dxl), ldxl, rows, cols, _dev(dxn), _stream(dxn)), "rx_cross_combine_bwd")
    return du, dxl


def cross_combine_bwd_stack(dxn, x0, u, dx0_acc, accumulate: bool, identity: bool, want_dxl: bool):
    """One layer of a cross stack walked backwards with a single dX0 buffer (see
    rx_cross_combine_bwd_stack).  Returns (du, dxl or None)."""
    dxn, ldd = _mat(dxn, "dxn"); x0, ld0 = _mat(x0, "x0"); u, ldu = _mat(u, "u")
    dx0_acc, ldx0 = _mat(dx0_acc, "dx0")
    rows, cols = dxn.shape
    du = _padded(rows, cols, dxn.device)
    _, lddu = _mat(du, "du")
    dxl, ldxl = None, 0
    if want_dxl:
        dxl = _padded(rows, cols, dxn.device)
        _, ldxl = _mat(dxl, "dxl")
    check(lib.rx_cross_combine_bwd_stack(_p(dxn), ldd, _p(x0), ld0, _p(u), ldu, _p(du), lddu, _p(dx0_acc), ldx0,
                                         _p(dxl), ldxl, rows, cols, int(accumulate), int(identity), _dev(dxn), _stream(dxn)),
          "rx_cross_combine_bwd_stack")
    return du, dxl


# ------------------------------------------------------------------ exchange helpers / eval
def bucketize_by_owner(idx: Tensor, world: int, n_rows_global: int, n_valid: Optional[Tensor] = None):
    """``n_valid``: device int32 count of the leading entries that are real (the rest is padding)."""
    _idx(idx, "indices")
    flat = idx.reshape(-1)
    n, dev = flat.numel(), idx.device
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    send_local = torch.empty(n, dtype=torch.int64, device=dev)
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    ws = _ws(int(lib.rx_bucketize_workspace_bytes(n, world)), dev)
    check(lib.rx_bucketize_by_owner(_p(flat), n, _p(n_valid), world, n_rows_global, _p(counts), _p(send_local), _p(perm),
                                    _p(ws), ws.numel(), _p(_err_flag(dev)), _dev(idx), _stream(idx)),
          "rx_bucketize_by_owner")
    return counts, send_local, perm


def unpermute_rows(rows: Tensor, perm: Tensor, n_valid: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """out[perm[j]] = rows[j] for the first ``n_valid`` (device count; default all) entries of ``perm``.
    ``out``: destination with at least max(perm) + 1 rows (default: same shape as ``rows``)."""
    if out is None:
        out = torch.empty_like(rows)
    dim = rows.shape[1] if rows.dim() == 2 else 1
    # every real entry owns a distinct row of ``out``, so at most out.shape[0] entries are real
    n = min(perm.numel(), out.shape[0]) if n_valid is not None else perm.numel()
    check(lib.rx_unpermute_rows(_p(rows), _p(perm), n, _p(n_valid), dim, _p(out), _dev(rows), _stream(rows)),
          "rx_unpermute_rows")
    return out


def permute_rows(rows: Tensor, perm: Tensor, n_valid: Optional[Tensor] = None) -> Tensor:
    """out[j] = rows[perm[j]] for the first ``n_valid`` entries of ``perm``; ``out`` has one row per entry."""
    dim = rows.shape[1] if rows.dim() == 2 else 1
    out = torch.empty((perm.numel(), dim) if rows.dim() == 2 else (perm.numel(),), dtype=rows.dtype, device=rows.device)
    n = min(perm.numel(), rows.shape[0]) if n_valid is not None else perm.numel()   # real entries read distinct rows
    check(lib.rx_permute_rows(_p(rows), _p(perm), n, _p(n_valid), dim, _p(out), _dev(rows), _stream(rows)),
          "rx_permute_rows")
    return out


def pack_by_owner(send_local: Tensor, perm: Tensor, counts: Tensor, capacity: int, pad_rows: int = 1,
                  out: Optional[Tuple[Tensor, Tensor]] = None):
    """Owner-grouped list -> fixed-capacity send buffer.  Returns (packed [G * capacity] int64,
    src_of [n] int32: packed slot of every list position).  Padding slot i names local row
    i % pad_rows (``pad_rows`` <= every owner's local height).  ``out``: a (packed, src_of) pair
    from an earlier call with the same sizes, overwritten in place."""
    G, n = counts.numel(), send_local.numel()
    if out is not None:
        packed, src_of = out
        if (packed.dtype != torch.int64 or src_of.dtype != torch.int32 or packed.numel() != G * capacity or src_of.numel() != n
                or not packed.is_contiguous() or not src_of.is_contiguous()):
            raise ValueError("pack_by_owner: `out` was made for other sizes")
    else:
        packed = torch.empty(G * capacity, dtype=torch.int64, device=send_local.device)
        src_of = torch.empty(n, dtype=torch.int32, device=send_local.device)
    check(lib.rx_pack_by_owner(_p(send_local), _p(perm), _p(counts), G, n, capacity, pad_rows, _p(packed), _p(src_of),
                               _p(_err_flag(send_local.device)), _dev(send_local), _stream(send_local)),
          "rx_pack_by_owner")
    return packed, src_of


def fm_flat_rows(sparse: Tensor, offsets: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """rx_fm_flat_rows: [B, fields] field-local indices + [fields] offsets -> the occurrence list [B * fields]
    (fields 0 .. fields-2 sample by sample, then the last field's B entries)."""
    _idx(sparse, "sparse"); _idx(offsets, "offsets")
    if sparse.dim() != 2 or offsets.numel() != sparse.shape[1] or not sparse.is_contiguous() or not offsets.is_contiguous():
        raise ValueError("fm_flat_rows: sparse [B, fields] and offsets [fields], contiguous")
    _same_device(sparse, offsets)
    B, fields = sparse.shape
    if out is None:
        out = torch.empty(B * fields, dtype=torch.int64, device=sparse.device)
    elif out.dtype != torch.int64 or out.numel() != B * fields or not out.is_contiguous() or out.device != sparse.device:
        raise ValueError("fm_flat_rows: `out` was made for a batch of another shape")
    check(lib.rx_fm_flat_rows(_p(sparse), _p(offsets), B, fields, _p(out), _dev(sparse), _stream(sparse)), "rx_fm_flat_rows")
    return out


def compose_slots(src_of: Tensor, inverse: Tensor, out: Optional[Tuple[Tensor, Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """rx_compose_slots: (slot_of [n_list] int64 = src_of, occ_slot [n_occ] int64 = src_of[inverse])."""
    if src_of.dtype != torch.int32 or inverse.dtype != torch.int32 or not src_of.is_contiguous() or not inverse.is_contiguous():
        raise ValueError("compose_slots: src_of and inverse must be contiguous int32 vectors")
    _same_device(src_of, inverse)
    n_list, n_occ = src_of.numel(), inverse.numel()
    if out is None:
        out = (torch.empty(n_list, dtype=torch.int64, device=src_of.device), torch.empty(n_occ, dtype=torch.int64, device=src_of.device))
    elif (out[0].dtype != torch.int64 or out[1].dtype != torch.int64 or out[0].numel() != n_list or out[1].numel() != n_occ
          or not out[0].is_contiguous() or not out[1].is_contiguous()):
        raise ValueError("compose_slots: `out` was made for lists of other lengths")
    check(lib.rx_compose_slots(_p(src_of), _p(inverse), n_list, n_occ, _p(out[0]), _p(out[1]), _dev(src_of), _stream(src_of)),
          "rx_compose_slots")
    return out


def zero_unused_slots(rows: Tensor, companion: Optional[Tensor], counts: Tensor, capacity: int) -> None:
    """rx_zero_unused_slots: zero the padding slots (offset >= counts[owner]) of a fixed-capacity send buffer
    ``rows`` [G * capacity, dim] and of ``companion`` [G * capacity] in place; used slots are left as they are."""
    _req(rows, torch.float32, "rows", 2)
    G = counts.numel()
    if counts.dtype != torch.int64 or not counts.is_contiguous() or rows.shape[0] != G * capacity or not rows.is_contiguous():
        raise ValueError("zero_unused_slots: counts [G] int64, rows [G * capacity, dim] contiguous")
    if companion is not None and (companion.dtype != torch.float32 or companion.numel() != rows.shape[0] or not companion.is_contiguous()):
        raise ValueError("zero_unused_slots: companion must be float32 [G * capacity]")
    check(lib.rx_zero_unused_slots(_p(counts), G, capacity, rows.shape[1], _p(rows), _p(companion), _dev(rows), _stream(rows)),
          "rx_zero_unused_slots")


TOPK_MAX_CANDIDATES = 1024   # widest score row rx_topk_codes takes (one wavefront per row, candidates in registers)


def topk_codes(scores: Tensor, codes: Optional[Tensor], k: int) -> Tensor:
    _req(scores, torch.float32, "scores", 2)
    if codes is not None:
        _idx(codes, "codes")
        if codes.shape != scores.shape:
            raise ValueError("codes and scores must have the same shape")
    rows, n = scores.shape
    out = torch.empty(rows, k, dtype=torch.int64, device=scores.device)
    check(lib.rx_topk_codes(_p(scores), _p(codes), rows, n, k, _p(out), _dev(scores), _stream(scores)),
          "rx_topk_codes")
    return out


# ------------------------------------------------------------------ batching
def sample_negatives(users: Tensor, items: Tensor, pool: Tensor, n_neg: int, seed: int,
                     want_labels: bool = True) -> Tuple[Tensor, Optional[Tensor]]:
    """rx_sample_negatives: ``[n, 1 + n_neg]`` item codes (the positive first, then ``n_neg`` uniform
    draws with replacement from ``pool[user]``) and the float64 ``[1, 0, ..., 0]`` label rows.
    The draws depend on (seed, position, draw number) only; ``sample_negatives_host`` is the same
    function in numpy."""
    _idx(users, "users"); _idx(items, "items")
    _req(pool, torch.int64, "pool", 2)
    n = users.numel()
    if users.dim() != 1 or items.shape != users.shape:
        raise ValueError("users and items: expected two [n] vectors")
    _same_device(users, items); _same_device(users, pool)
    dev = users.device
    out = torch.empty(n, 1 + n_neg, dtype=torch.int64, device=dev)
    labels = torch.empty(n, 1 + n_neg, dtype=torch.float64, device=dev) if want_labels else None
    check(lib.rx_sample_negatives(_p(users), _p(items), _p(pool), n, pool.shape[0], pool.shape[1], n_neg,
                                  int(seed) & 0xFFFFFFFFFFFFFFFF, _p(out), _p(labels), _p(_err_flag(dev)), _dev(users),
                                  _stream(users)), "rx_sample_negatives")
    _after_index_op(dev)
    return out, labels


def epoch_advance(epoch: Dict[str, Tensor], batch: Dict[str, Tensor], n_batches: int, cursor: Tensor, ticket: Tensor,
                  loss: Optional[Tensor] = None, running: Optional[Tensor] = None, ahead: int = 1) -> None:
    """rx_epoch_advance: ``running += loss``, then batch ``cursor + ahead`` of every tensor of ``epoch``
    (slices of ``batch[k].shape[0]`` rows along dim 0) goes into ``batch[k]``, and the cursor moves on
    by one.  Meant to be recorded at the end of a graphed train step (utils.LoopGraph; ``ahead`` = 2
    when two static batches take turns)."""
    from ._lib import BatchPlan, RX_BATCH_MAX_TENSORS
    keys = list(batch)
    if not 1 <= len(keys) <= RX_BATCH_MAX_TENSORS:
        raise ValueError(f"1..{RX_BATCH_MAX_TENSORS} tensors per batch")
    _req(cursor, torch.int64, "cursor"); _req(ticket, torch.int32, "ticket")
    plan = BatchPlan()
    plan.count = len(keys)
    rows = batch[keys[0]].shape[0]
    plan.rows_per_batch, plan.n_batches, plan.ahead = rows, int(n_batches), int(ahead)
    for t, k in enumerate(keys):
        src, dst = epoch[k], batch[k]
        if not (src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype
                and src.shape[1:] == dst.shape[1:] and dst.shape[0] == rows and src.shape[0] >= n_batches * rows):
            raise ValueError(f"epoch_advance: tensor {k!r} does not match its epoch tensor")
        plan.src[t], plan.dst[t] = src.data_ptr(), dst.data_ptr()
        plan.row_bytes[t] = dst[0].numel() * dst.element_size() if dst.dim() > 1 else dst.element_size()
    if (loss is None) != (running is None):
        raise ValueError("epoch_advance: loss and running come as a pair")
    if loss is not None:
        _req(running, torch.float64, "running")
        if loss.dtype not in (torch.float64, torch.float32) or loss.numel() != 1:
            raise ValueError("epoch_advance: the loss is one float or double")
    dev = cursor.device
    check(lib.rx_epoch_advance(C.byref(plan), _p(cursor), _p(ticket), _p(loss),
                               int(loss is not None and loss.dtype == torch.float64), _p(running),
                               dev.index if dev.index is not None else torch.cuda.current_device(),
                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "rx_epoch_advance")


def sample_negatives_host(users, items, pool, n_neg: int, seed: int):
    """The draw rule of rx_sample_negatives in numpy (uint64 arithmetic wraps like the device's)."""
    import numpy as np
    users, items, pool = (np.asarray(x) for x in (users, items, pool))
    n, P = len(users), pool.shape[1]
    out = np.empty((n, 1 + n_neg), dtype=np.int64)
    out[:, 0] = items
    if n_neg:
        with np.errstate(over="ignore"):
            c = (np.arange(n, dtype=np.uint64)[:, None] * np.uint64(n_neg) + np.arange(n_neg, dtype=np.uint64)[None, :])
            z = np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF) + (c + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            r = ((z >> np.uint64(32)) * np.uint64(P)) >> np.uint64(32)
        out[:, 1:] = pool[users[:, None], r.astype(np.int64)]
    return out


# ------------------------------------------------------------------ model-level pieces of the backend surface
# (the sharded models take their kernels from a backend module; these two complete that surface so
# that a model never asks which backend it runs on)
def bce_loss(logits: Tensor, labels: Tensor) -> Tensor:
    """Mean BCE-with-logits with the reference's fp64 promotion, as an autograd node (rx_bce_fwd)."""
    from .autograd import bce_with_logits
    return bce_with_logits(logits, labels)


def run_tower(seq, x: Tensor, bf16: bool = False) -> Tensor:
    """An nn.Sequential tower (Linear / BatchNorm1d / ReLU / Dropout) through the dense-layer kernels."""
    from .models.base import run_tower as _run
    return _run(seq, x, bf16=bf16)


# ------------------------------------------------------------------ dense layers with bf16 activations in HBM
def _mat16(t: Tensor, name: str) -> Tuple[Tensor, int]:
    """2-D bfloat16 with unit inner stride and 16-byte addressable rows."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 2):
        raise TypeError(f"{name}: expected a 2-D bfloat16 GPU tensor")
    if t.stride(1) != 1 and t.shape[1] != 1:
        raise ValueError(f"{name}: inner stride must be 1")
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    return t, ld


def _padded16(rows: int, cols: int, device) -> Tensor:
    """bf16 [rows, cols] whose row stride is a multiple of 8 elements (16 bytes); padding columns are
    zero and stay zero (the kernels read and write whole 16-byte pieces)."""
    cp = (cols + 7) // 8 * 8
    if cp == cols:
        return torch.empty(rows, cp, dtype=torch.bfloat16, device=device)
    return torch.zeros(rows, cp, dtype=torch.bfloat16, device=device)[:, :cols]


def to_bf16(x: Tensor, transpose: bool = False) -> Tensor:
    """rx_to_bf16: round an fp32 matrix to bf16 (nearest even); ``transpose`` for weight matrices."""
    x, ldx = _mat(x, "x")
    rows, cols = x.shape
    out = _padded16(cols, rows, x.device) if transpose else _padded16(rows, cols, x.device)
    _, ldo = _mat16(out, "out")
    check(lib.rx_to_bf16(_p(x), ldx, _p(out), ldo, rows, cols, int(transpose), _dev(x), _stream(x)), "rx_to_bf16")
    return out


def linear16_fwd(x16: Tensor, w: Tensor, b: Optional[Tensor], relu: bool, out_bf16: bool, w16: Optional[Tensor] = None) -> Tensor:
    """act(x16 w^T + b) from bf16 activations; ``w`` the fp32 weight [N, K] (its bf16 copy is made
    here unless ``w16`` is passed).  Output bf16 (feeds the next layer) or fp32."""
    x16, ldx = _mat16(x16, "x16")
    _req(w, torch.float32, "weight", 2)
    M, K = x16.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"linear16: x is [{M},{K}] but weight is {tuple(w.shape)}")
    if b is not None:
        _req(b, torch.float32, "bias", 1)
    if N > 1 or out_bf16:
        w16 = w16 if w16 is not None else to_bf16(w)
    out = _padded16(M, N, x16.device) if out_bf16 else _padded(M, N, x16.device)
    ldy = out.stride(0) if M > 1 else max(N, out.stride(0))
    check(lib.rx_linear16_fwd(_p(x16), ldx, _p(w16), _p(w), _p(b), _p(out), ldy, M, N, K, int(relu), int(out_bf16),
                              _dev(x16), _stream(x16)), "rx_linear16_fwd")
    return out


def linear16_bwd_x(dy16: Tensor, w: Tensor, gate16: Optional[Tensor] = None, out_bf16: bool = False,
                   out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """dX = dY16 W; bf16 output with the ReLU gate of the layer in front (``gate16`` = its bf16
    output) or fp32 output, optionally accumulated into ``out``."""
    dy16, ldd = _mat16(dy16, "dy16")
    _req(w, torch.float32, "weight", 2)
    M, N = dy16.shape
    K = w.shape[1]
    w16t = to_bf16(w, transpose=True) if N > 1 else None
    ldg = 0
    if gate16 is not None:
        gate16, ldg = _mat16(gate16, "gate16")
    if out is None:
        out = _padded16(M, K, dy16.device) if out_bf16 else _padded(M, K, dy16.device)
    ldx = out.stride(0) if M > 1 else max(K, out.stride(0))
    check(lib.rx_linear16_bwd_x(_p(dy16), ldd, _p(w16t), _p(w), _p(out), ldx, _p(gate16), ldg, M, N, K, int(accumulate),
                                int(out_bf16), _dev(dy16), _stream(dy16)), "rx_linear16_bwd_x")
    return out


def linear16_bwd_w(dy16: Tensor, x16: Tensor, want_bias: bool = True):
    dy16, ldd = _mat16(dy16, "dy16")
    x16, ldx = _mat16(x16, "x16")
    M, N = dy16.shape
    K = x16.shape[1]
    dw = torch.empty(N, K, dtype=torch.float32, device=dy16.device)
    db = torch.empty(N, dtype=torch.float32, device=dy16.device) if want_bias else None
    ws = _ws(int(lib.rx_linear16_bwd_w_workspace_bytes(M, N, K)), dy16.device)
    check(lib.rx_linear16_bwd_w(_p(dy16), ldd, _p(x16), ldx, _p(dw), _p(db), M, N, K, _p(ws), ws.numel(), _dev(dy16),
                                _stream(dy16)), "rx_linear16_bwd_w")
    return dw, db


def cross16_fwd(x0, u, xl):
    """x_next = x0 * u + x_l: (fp32, its bf16 copy)."""
    x0, ld0 = _mat(x0, "x0"); u, ldu = _mat(u, "u"); xl, ldl = _mat(xl, "xl")
    rows, cols = x0.shape
    xn = _padded(rows, cols, x0.device)
    xn16 = _padded16(rows, cols, x0.device)
    check(lib.rx_cross16_fwd(_p(x0), ld0, _p(u), ldu, _p(xl), ldl, _p(xn), xn.stride(0), _p(xn16), xn16.stride(0), rows, cols,
                             _dev(x0), _stream(x0)), "rx_cross16_fwd")
    return xn, xn16


def cross16_layer_fwd(x16: Tensor, w: Tensor, b: Tensor, x0: Tensor, xl: Tensor):
    """One cross layer in one launch: returns (u, x_next, x_next16) with u = x16 w^T + b and
    x_next = x0 * u + x_l (rx_cross16_layer_fwd)."""
    x16, ldx = _mat16(x16, "x16")
    x0, ld0 = _mat(x0, "x0"); xl, ldl = _mat(xl, "xl")
    _req(w, torch.float32, "weight", 2); _req(b, torch.float32, "bias", 1)
    M, d = x16.shape
    if tuple(w.shape) != (d, d) or tuple(x0.shape) != (M, d) or tuple(xl.shape) != (M, d):
        raise ValueError("cross16_layer_fwd: a cross layer is square and x0 / x_l have the input's shape")
    w16 = to_bf16(w)
    u, xn, xn16 = _padded(M, d, x16.device), _padded(M, d, x16.device), _padded16(M, d, x16.device)
    check(lib.rx_cross16_layer_fwd(_p(x16), ldx, _p(w16), _p(b), _p(x0), ld0, _p(xl), ldl, _p(u), u.stride(0), _p(xn), xn.stride(0),
                                   _p(xn16), xn16.stride(0), M, d, _dev(x16), _stream(x16)), "rx_cross16_layer_fwd")
    return u, xn, xn16


def cross16_bwd(dxn, x0, u, dx0_acc, accumulate: bool, identity: bool, want_dxl: bool):
    """cross_combine_bwd_stack with du written as bf16 only.  Returns (du16, dxl or None)."""
    dxn, ldd = _mat(dxn, "dxn"); x0, ld0 = _mat(x0, "x0"); u, ldu = _mat(u, "u")
    dx0_acc, ldx0 = _mat(dx0_acc, "dx0")
    rows, cols = dxn.shape
    du16 = _padded16(rows, cols, dxn.device)
    dxl, ldxl = None, 0
    if want_dxl:
        dxl = _padded(rows, cols, dxn.device)
        ldxl = dxl.stride(0)
    check(lib.rx_cross16_bwd(_p(dxn), ldd, _p(x0), ld0, _p(u), ldu, _p(du16), du16.stride(0), _p(dx0_acc), ldx0, _p(dxl), ldxl,
                             rows, cols, int(accumulate), int(identity), _dev(dxn), _stream(dxn)), "rx_cross16_bwd")
    return du16, dxl


def bn_apply16(x: Tensor, mean: Tensor, rstd: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], relu: bool,
               keep: Optional[Tensor] = None, keep_scale: float = 1.0) -> Tensor:
    """rx_bn_apply16: the normalised (+ ReLU, + dropout) output as bf16 only."""
    x, ldx = _mat(x, "x")
    M, C = x.shape
    if keep is not None and not (keep.dtype == torch.uint8 and keep.is_contiguous() and tuple(keep.shape) == (M, C)):
        raise ValueError("keep: expected a contiguous uint8 [M, C] mask")
    out = _padded16(M, C, x.device)
    check(lib.rx_bn_apply16(_p(x), ldx, M, C, _p(mean), _p(rstd), _p(gamma), _p(beta), int(relu), _p(keep), keep_scale,
                            _p(out), out.stride(0) if M > 1 else max(C, out.stride(0)), _dev(x), _stream(x)), "rx_bn_apply16")
    return out


def bn_bwd16(dout: Tensor, x: Tensor, out16: Optional[Tensor], mean: Tensor, rstd: Tensor, gamma: Optional[Tensor],
             gate: bool, keep_scale: float, batch_stats: bool):
    """rx_bn_bwd16: (dx as bf16, dgamma, dbeta); ``out16`` is the bf16 output of bn_apply16 (the gate)."""
    dout, ldd = _mat(dout, "dout")
    x, ldx = _mat(x, "x")
    M, C = x.shape
    ldo = 0
    if gate:
        out16, ldo = _mat16(out16, "out16")
    dx = _padded16(M, C, x.device)
    dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = _ws(int(lib.rx_bn_workspace_bytes(M, C)), x.device)
    check(lib.rx_bn_bwd16(_p(dout), ldd, _p(x), ldx, _p(out16) if gate else None, ldo, M, C, _p(mean), _p(rstd), _p(gamma),
                          int(gate), keep_scale, int(batch_stats), _p(dx), dx.stride(0) if M > 1 else max(C, dx.stride(0)),
                          _p(dgamma), _p(dbeta), _p(ws), ws.numel(), _dev(x), _stream(x)), "rx_bn_bwd16")
    return dx, dgamma, dbeta
