"""torch.autograd glue: each Function's forward and backward is one or a few C-ABI launches.

Embedding-table gradients never go through aten's embedding_dense_backward.  Each lookup site
yields per-occurrence gradient rows; ``table_backward`` sorts/dedups the indices
(rx_sort_dedup) and runs the segmented reduction (rx_segreduce) which either

* writes a dense ``[rows, dim]`` gradient (drop-in mode: any torch optimizer, clip_grad_norm_
  and the reference train loop keep working unchanged), or
* applies the sparse row update in the same pass when the parameter carries a sparse rule
  (see ``optim.FusedOptimizer``); the parameter's ``.grad`` then stays ``None``.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
from torch.autograd import Function

from . import _lib, ops
import os

from .ops import FieldGroups, Segment, SortedIndex

Tensor = torch.Tensor


_SORT_CACHE: list = []
_PRESORTED: dict = {}


def _sort_key(flat: Tensor, n_rows: int):
    return (flat.data_ptr(), flat.numel(), flat._version, int(n_rows), flat.device)


def presort(idx: Tensor, n_rows: int, out=None):
    """Sort + dedup an index tensor ahead of the step that will need it (``HotPathModule.prepare``;
    typically on a side stream while the previous step runs) and pin the result so that the
    backward's ``sorted_index`` finds it.  ``out``: a SortedIndex from an earlier call for the same
    tensor, overwritten in place (static buffers for graph replay)."""
    flat = idx.reshape(-1)
    if not flat.is_contiguous():
        raise ValueError("presort: the index tensor must be contiguous")
    si = ops.sort_dedup(flat, n_rows, out=out)
    _PRESORTED[_sort_key(flat, n_rows)] = (flat, si)
    while len(_PRESORTED) > 512:
        _PRESORTED.pop(next(iter(_PRESORTED)))
    return si


_CAPTURE_CACHE: list = []


def clear_sort_cache() -> None:
    """Forget every memoised sort.  ``graph.GraphedTrainStep`` calls this before it records a step, so
    that the recording contains its own sort kernels writing to buffers of the graph's pool."""
    _SORT_CACHE.clear()
    _CAPTURE_CACHE.clear()


def clear_capture_sort_cache() -> None:
    """Forget the sorts memoised during the current recording (between the steps of a multi-step one)."""
    _CAPTURE_CACHE.clear()


def sorted_index(idx: Tensor, n_rows: int):
    """``ops.sort_dedup`` with a small memo: NeuMF-style models look the same index tensor up in
    several tables of equal height (GMF and MLP embeddings), and one sort serves them all.  An
    entry is keyed by storage address, length and version counter and keeps its index tensor
    alive, so a hit can only be the same, unmodified data.  Results pinned by ``presort`` are
    found first.

    While a stream is being captured into a HIP graph the eager memo is neither read nor written: a
    hit there would leave the sort out of the recording, and a replay after the indices were
    refreshed in place would scatter gradients with the old batch's order.  A separate memo, emptied
    by ``clear_sort_cache`` before each recording, still shares one sort between the tables of one
    recorded step."""
    flat = idx.reshape(-1)
    key = _sort_key(flat, n_rows)
    capturing = flat.is_cuda and torch.cuda.is_current_stream_capturing()
    hit = _PRESORTED.get(key)
    if hit is not None:
        return hit[1]
    cache = _CAPTURE_CACHE if capturing else _SORT_CACHE
    for k, _, si, made in cache:
        if k == key:
            _follow(made, si)
            return si
    si = ops.sort_dedup(flat.contiguous() if not flat.is_contiguous() else flat, n_rows)
    if flat.is_contiguous():
        made = None
        if flat.is_cuda:      # a model whose branches run on two streams may ask from the other one
            made = (torch.cuda.current_stream(flat.device), torch.cuda.Event())
            made[1].record(made[0])
        cache.append((key, flat, si, made))
        if len(cache) > (64 if capturing else 8):
            cache.pop(0)
    return si


def _follow(made, si) -> None:
    """A memoised sort is about to be read on the current stream: if another stream produced it, wait
    for that launch and tell the allocator about the second reader."""
    if made is None:
        return
    cur = torch.cuda.current_stream(si.seg_off.device)
    if cur != made[0]:
        cur.wait_event(made[1])
        for t in (si.sorted_pos, si.sorted_slot, si.uniq_rows, si.seg_off, si.n_uniq, si.inverse):
            if t is not None:
                t.record_stream(cur)


def table_backward(param: Optional[Tensor], si: SortedIndex, occ: Tensor, dim: Optional[int] = None,
                   occ_ld: Optional[int] = None, param2: Optional[Tensor] = None,
                   occ2: Optional[Tensor] = None):
    """Reduce per-occurrence gradients ``occ`` (rows ordered like the index list behind ``si``)
    into ``param``'s gradient, or apply the parameter's fused sparse rule.  ``param2`` / ``occ2``:
    a companion [rows, 1] table gathered with the same indices, handled in the same launches.
    Returns the dense gradient (or None), or a pair when a companion is given."""
    if param is None:
        return None if param2 is None else (None, None)
    rule = getattr(param, "_rx_sparse", None)
    if rule is not None:
        rule.apply(param, occ, si, dim=dim, occ_ld=occ_ld, param2=param2, occ2=occ2)
        return None if param2 is None else (None, None)
    grad = torch.zeros_like(param)
    grad2 = torch.zeros_like(param2) if param2 is not None else None
    ops.segreduce(occ, si, _lib.RX_SEG_DENSE, grad, dim=dim, occ_ld=occ_ld, occ2=occ2, target2=grad2)
    return grad if param2 is None else (grad, grad2)


def tables_backward(params: Sequence[Tensor], sis: Sequence[SortedIndex], occs: Sequence[Tensor]):
    """``table_backward`` for several tables at once.  Tables without a fused sparse rule, all of one width on
    one GPU, get their dense gradients from ONE zeroed buffer and ONE pair of reduce launches
    (rx_segreduce_multi in DENSE mode: the same bits as one reduce per table); anything else goes table by
    table."""
    same = (len(params) > 1 and all(p is not None and p.is_cuda and getattr(p, "_rx_sparse", None) is None and p.dim() == 2
                                    and p.shape[1] == params[0].shape[1] and p.dtype == torch.float32 for p in params)
            and len(params) <= _lib.RX_SEG_MULTI_MAX and os.environ.get("RX_NO_MULTI_TABLE") != "1")
    if not same:
        return [table_backward(p, si, occ) for p, si, occ in zip(params, sis, occs)]
    sizes = [(p.numel() + 63) // 64 * 64 for p in params]         # every gradient starts on a 256-byte boundary
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=params[0].device)
    grads, o = [], 0
    for p, n in zip(params, sizes):
        grads.append(flat[o:o + p.numel()].view(p.shape))
        o += n
    ops.segreduce_multi([ops.TableReduce(occ, si, g, p.shape[1]) for p, si, occ, g in zip(params, sis, occs, grads)], _lib.RX_SEG_DENSE)
    return grads


# ------------------------------------------------------------------ MF-family score
class MFScore(Function):
    """logits[B,N] of the matrix-factorization family / GMF (rx_mf_score_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, users, items, occs, temps, U, V, O, UT, T, user_bias, item_bias, bias, lin_w, lin_b):
        kw = dict(O=O, occs=occs, UT=UT, T=T, temps=temps, user_bias=user_bias, item_bias=item_bias,
                  bias=bias, lin_w=None if lin_w is None else lin_w.reshape(-1), lin_b=lin_b)
        ctx.kw = kw
        ctx.args = (U, V, users, items)
        ctx.lin_w = lin_w
        return ops.mf_score_fwd(U, V, users, items, **kw)

    @staticmethod
    def backward(ctx, g):
        return _mf_backward(ctx, g.contiguous())


def _mf_backward(ctx, g):
        """Gradients of everything an MF-family score reads, from dL/dlogit ``g`` [B, N]."""
        U, V, users, items = ctx.args
        kw = ctx.kw
        user_occ, item_occ, utemp_occ, temp_occ, gsum, linw_grad = ops.mf_score_bwd(g, U, V, users, items, loss_final=getattr(ctx, "loss_final", None), **kw)
        ctx.loss_final = None
        su = sorted_index(users, U.shape[0])
        si = sorted_index(items, V.shape[0])
        gU, gV = tables_backward((U, V), (su, si), (user_occ, item_occ))
        gO = gUT = gT = gub = gib = gb = glw = glb = None
        if kw["O"] is not None:
            gO = table_backward(kw["O"], sorted_index(kw["occs"], kw["O"].shape[0]), user_occ)
        if kw["UT"] is not None:
            gUT = table_backward(kw["UT"], su, utemp_occ)
            gT = table_backward(kw["T"], sorted_index(kw["temps"], kw["T"].shape[0]), temp_occ)
        if kw["user_bias"] is not None:
            gub = table_backward(kw["user_bias"], su, gsum, dim=1)
        if kw["item_bias"] is not None:
            gib = table_backward(kw["item_bias"], si, g.reshape(-1), dim=1)
        if kw["bias"] is not None:
            gb = ops.reduce_sum(g.reshape(-1)).reshape(kw["bias"].shape)
        if ctx.lin_w is not None:
            glw = linw_grad.reshape(ctx.lin_w.shape)
        if kw["lin_b"] is not None:
            glb = ops.reduce_sum(g.reshape(-1)).reshape(kw["lin_b"].shape)
        return (None, None, None, None, gU, gV, gO, gUT, gT, gub, gib, gb, glw, glb)


class MFLoss(Function):
    """Forward + BCE loss of the MF family / GMF as one launch (rx_mf_score_fwd_bce): the forward leaves
    dL/dlogit, so the backward starts at rx_mf_score_bwd with no loss kernel in between.  Same arguments
    as MFScore plus the labels [B, N]."""

    @staticmethod
    def forward(ctx, users, items, occs, temps, U, V, O, UT, T, user_bias, item_bias, bias, lin_w, lin_b, labels):
        kw = dict(O=O, occs=occs, UT=UT, T=T, temps=temps, user_bias=user_bias, item_bias=item_bias,
                  bias=bias, lin_w=None if lin_w is None else lin_w.reshape(-1), lin_b=lin_b)
        ctx.kw = kw
        ctx.args = (U, V, users, items)
        ctx.lin_w = lin_w
        # inside a train step (DEFER_LOSS_FINAL, set by the step drivers: a backward pass is certain to follow) the
        # launch that adds the loss partials is left to the backward launch
        defer = DEFER_LOSS_FINAL[0] and labels.dtype == torch.float64 and users.shape[0] > 0 and any(ctx.needs_input_grad[4:14])
        out = ops.mf_score_fwd_bce(U, V, users, items, labels.contiguous(), defer_final=defer, **kw)
        loss, ctx.dlogits = out[0], out[1]
        ctx.loss_final = out[3] if defer else None
        return loss if labels.dtype == torch.float64 else loss.to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        d = ctx.dlogits
        if not _UNIT_LOSS_GRAD:   # an upstream factor other than 1 (loss * c): scale the saved gradient
            d = d * g.to(torch.float32)
        return _mf_backward(ctx, d) + (None,)


# ------------------------------------------------------------------ FM second order (+ DeepFM tower input)
_side_streams = {}


def _side_stream(device) -> "torch.cuda.Stream":
    key = torch.device(device).index
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


_BRANCHES_OFF = os.environ.get("RX_NO_BRANCH_STREAMS") == "1"      # A/B timing: every branch on the caller's stream
_open_branches: list = []
_branch_warning_off = [False]


class Branch:
    """``with Branch(device) as br:`` runs the launches of an independent branch of a model (NeuMF's
    GMF half beside its MLP tower) on a second stream, forked from the current stream on entry;
    ``join`` brings the two together again.  Autograd runs a node's backward on the stream its
    forward ran on, so the branch's backward overlaps the other branch's as well.  What the branch's
    backward leaves for later (the per-occurrence gradients a clipped FusedOptimizer.step() reduces)
    is joined by ``join_branches``."""

    def __init__(self, device):
        self.on = torch.device(device).type == "cuda" and not _BRANCHES_OFF
        self.device = device

    def __enter__(self):
        if self.on:
            if not _branch_warning_off[0]:
                # a parameter used inside the branch gets its gradient on the branch's stream; the engine
                # syncs that correctly and warns about the mismatch with the stream of the accumulation
                # node, which is the intended state here
                off = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
                if off is not None:
                    off(False)
                _branch_warning_off[0] = True
            self.main = torch.cuda.current_stream(self.device)
            self.side = _side_stream_for("branch", self.device)
            self.side.wait_stream(self.main)
            self._ctx = torch.cuda.stream(self.side)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self._ctx.__exit__(*exc)
            if self.side not in _open_branches:
                _open_branches.append(self.side)
        return False

    def join(self, *tensors: Tensor) -> None:
        """Call where the caller's stream first needs the branch's results (after it has been given
        its own work): it waits for the branch there, and ``tensors`` made inside the branch are
        marked as read by the caller's stream."""
        if self.on:
            self.main.wait_stream(self.side)
            for t in tensors:
                t.record_stream(self.main)


def join_branches(device) -> None:
    """The current stream waits for everything the branch streams have been given (their share of
    the backward pass included).  FusedOptimizer.step() and backward_from_loss call it."""
    if not _open_branches:
        return
    cur = torch.cuda.current_stream(device)
    for s in _open_branches:
        if s.device == cur.device and s != cur:
            cur.wait_stream(s)
    _open_branches[:] = [s for s in _open_branches if s.device != cur.device]


def _side_stream_for(what: str, device) -> "torch.cuda.Stream":
    key = (what, torch.device(device).index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


_FG_MAX_DIM = int(os.environ.get("RX_FG_MAX_DIM", 32))


def use_field_groups(feat, ex_idx, item_idx, field_base) -> bool:
    """The field-grouped backward (rx_fg_*) serves one-item-per-sample batches over disjoint field
    ranges; ``RX_NO_FIELD_GROUPS=1`` keeps everything on the general sort (for A/B timing)."""
    if field_base is None or os.environ.get("RX_NO_FIELD_GROUPS") == "1":
        return False
    # Rows of up to 32 floats only.  Measured on the 33 M-row Criteo-shaped FM at B = 65536 (ms per step,
    # field groups / general sort): D = 16 0.153 / 0.197, D = 32 0.207 / 0.213, D = 64 0.398 / 0.332,
    # D = 128 0.943 / 0.570 -- the position-parallel reduce re-reads a D-wide record per occurrence and
    # loses to the run-walking one once rows are wide.
    if feat.shape[1] > _FG_MAX_DIM:
        return False
    return ops.fg_supported(ex_idx, item_idx, field_base, feat.shape[1])


def index_prep(feat, ex_idx, ex_base, item_idx, item_base, field_base=None, out=None):
    """Index work of an FM batch's backward: the field-grouped order when the batch qualifies
    (``field_base``: host row offsets of the fields + table height), else the general sort + dedup."""
    if use_field_groups(feat, ex_idx, item_idx, field_base):
        return ops.fg_group(ex_idx, item_idx, field_base, out=out if isinstance(out, FieldGroups) else None)
    return ops.fm_sort_dedup(feat, ex_idx, ex_base, item_idx, item_base, out=out if isinstance(out, SortedIndex) else None)


class _AsyncSort:
    """Index preparation for the backward (flat row ids -> sort -> dedup) depends on the batch's
    indices only, so it is issued on a side HIP stream at forward time and overlaps the forward,
    loss and interaction-backward kernels; the backward joins the stream before it reduces."""

    def __init__(self, feat, ex_idx, ex_base, item_idx, item_base, field_base=None):
        dev = feat.device
        self.main = torch.cuda.current_stream(dev)
        self.side = _side_stream(dev)
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side):
            self.si = index_prep(feat, ex_idx, ex_base, item_idx, item_base, field_base)

    def join(self):
        main = torch.cuda.current_stream(self.si.srow.device if isinstance(self.si, FieldGroups) else self.si.seg_off.device)
        main.wait_stream(self.side)
        if isinstance(self.si, FieldGroups):
            held = (self.si.srow, self.si.sb)
        else:
            held = (self.si.sorted_pos, self.si.sorted_slot, self.si.uniq_rows, self.si.seg_off, self.si.n_uniq)
        for t in held:
            t.record_stream(main)
        return self.si


class Presort:
    """Index preparation done ahead of the step (see HotPathModule.prepare): the sort / dedup
    result of a batch's row list, in buffers that can be refreshed in place."""

    def __init__(self, si):
        self.si = si

    def join(self):
        return self.si


def table_backward_fm(feat, bias_table, si, tsum, g, n_fields_ex: int, deep=None, premultiplied: bool = False):
    """N == 1 FM backward without per-occurrence rows (rx_segreduce_fm, or rx_fg_reduce when ``si``
    is a field-grouped list); ``deep``: gradient of the tower input, whose per-field slices join the
    rows' gradients in the same pass."""
    rule = getattr(feat, "_rx_sparse", None)
    if premultiplied and not isinstance(si, FieldGroups):
        raise RuntimeError("premultiplied rows are read by the field-grouped reduce only")
    if rule is not None:
        rule.apply_fm(feat, bias_table, si, tsum, g, n_fields_ex, deep, premultiplied)
        return None, None
    # dense gradients are back-to-back rows whatever the strides of the tables
    grad = torch.zeros(feat.shape, dtype=feat.dtype, device=feat.device)
    grad2 = torch.zeros(bias_table.shape, dtype=bias_table.dtype, device=bias_table.device) if bias_table is not None else None
    if isinstance(si, FieldGroups):
        ops.fg_reduce(tsum, g, si, _lib.RX_SEG_DENSE, grad, table=feat.detach(), target2=grad2, deep=deep,
                      premultiplied=premultiplied)
        return grad, grad2
    ops.segreduce_fm(tsum, g, n_fields_ex, si, _lib.RX_SEG_DENSE, grad, table=feat.detach(), target2=grad2, deep=deep)
    return grad, grad2


def _before_table_read(param) -> None:
    if getattr(param, "_rx_sparse", None) is not None:
        from .optim import before_table_read
        before_table_read(param)


class FM2(Function):
    """FM bias + second-order term over one shared table (rx_fm2_fwd / _bwd).  Returns
    ``(logits[B,N] or None, deep[B*N, deep_cols] or None)``.  ``deep`` holds the gathered rows
    [item, field 0..F-1] in its first (1+F)*D columns; further columns (``deep_cols`` wider)
    hold ``extra`` (dense features, which get no gradient)."""

    @staticmethod
    def forward(ctx, ex_idx, ex_base, item_idx, item_base, feat, bias_table, want_deep, second_order=True,
                deep_cols=None, want_logits=True, extra=None, presort=None, field_base=None):
        _before_table_read(feat)
        bias_flat = None if bias_table is None else bias_table.reshape(-1)
        n_items = item_idx.shape[1] if item_idx.dim() == 2 else 1
        # with one item per sample the backward needs only T_b = sum of the sample's rows (plus the
        # tower-input gradient when there is a tower): no per-occurrence gradient rows
        # (a tower-only model, e.g. DCN-v2, has no FM term: its rows receive the tower gradient alone)
        fused = n_items == 1 and ((second_order and (want_logits or want_deep)) or (want_deep and not want_logits))
        out = ops.fm2_fwd(feat, bias_flat, ex_idx, ex_base, item_idx, item_base, want_deep, second_order, deep_cols,
                          want_logits, extra, want_tsum=fused)
        ctx.args = (ex_idx, ex_base, item_idx, item_base, feat, bias_table)
        ctx.second_order = second_order
        ctx.tsum = out[2] if fused else None
        ctx.sort = presort
        ctx.field_base = field_base if fused else None   # the field-grouped order serves the fused backward only
        if presort is not None and isinstance(presort.join(), FieldGroups) and not fused:
            ctx.sort = None
        if ctx.sort is None and ctx.needs_input_grad[4] and feat.is_cuda:
            ctx.sort = _AsyncSort(feat, ex_idx, ex_base, item_idx, item_base, ctx.field_base)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g, g_deep):
        ex_idx, ex_base, item_idx, item_base, feat, bias_table = ctx.args
        if g is not None:
            g = g.contiguous()
        if g_deep is not None and g_deep.stride(-1) != 1:
            g_deep = g_deep.contiguous()
        none_tail = (None,) * 7
        if ctx.tsum is not None and (g is not None or g_deep is not None):
            s = ctx.sort.join() if ctx.sort is not None else index_prep(feat, ex_idx, ex_base, item_idx, item_base, ctx.field_base)
            gl = g.reshape(-1) if g is not None else torch.zeros(ctx.tsum.shape[0], dtype=torch.float32, device=feat.device)
            gfeat, gbias = table_backward_fm(feat, bias_table, s, ctx.tsum, gl, ex_idx.shape[1], g_deep)
            return (None, None, None, None, gfeat, gbias) + none_tail
        bias_flat = None if bias_table is None else bias_table.reshape(-1)
        occ, bias_occ = ops.fm2_bwd(g, g_deep, feat, bias_flat, ex_idx, ex_base, item_idx, item_base,
                                    ctx.second_order)
        s = ctx.sort.join() if ctx.sort is not None else ops.fm_sort_dedup(feat, ex_idx, ex_base, item_idx, item_base)
        if bias_table is not None:
            gfeat, gbias = table_backward(feat, s, occ, param2=bias_table, occ2=bias_occ)
        else:
            gfeat, gbias = table_backward(feat, s, occ), None
        return (None, None, None, None, gfeat, gbias) + none_tail


DEFER_LOSS_FINAL = [False]     # see FMLoss.forward; switched on by graph.GraphedTrainStep._eager and utils.train_step


class FMLoss(Function):
    """Forward + BCE loss of the one-item-per-sample FM as one launch (rx_fm2_fwd_bce), backward over
    the field-grouped occurrence list.  Used by FieldFM.loss when the batch qualifies for field
    groups; the forward leaves g_b * T_b and g_b per sample, so the backward starts at the reduce."""

    @staticmethod
    def forward(ctx, ex_idx, ex_base, item_idx, item_base, feat, bias_table, labels, presort, field_base):
        _before_table_read(feat)
        bias_flat = None if bias_table is None else bias_table.reshape(-1)
        # inside a train step (DEFER_LOSS_FINAL, set by the step drivers: a backward pass is certain to follow) the
        # launch that adds the loss partials is left out and the backward's reduce writes the loss instead
        defer = DEFER_LOSS_FINAL[0] and ctx.needs_input_grad[4] and labels.dtype == torch.float64     # (grad mode is off inside forward(): needs_input_grad is the test)
        out = ops.fm2_fwd_bce(feat, bias_flat, ex_idx, ex_base, item_idx, item_base, labels.contiguous(), defer_final=defer)
        loss, gt = out[0], out[1]
        ctx.loss_final = out[3] if defer else None
        ctx.args = (ex_idx, ex_base, item_idx, item_base, feat, bias_table)
        ctx.gt, ctx.field_base = gt, field_base
        ctx.sort = presort
        if presort is not None and not isinstance(presort.join(), FieldGroups):
            ctx.sort = None
        if ctx.sort is None and ctx.needs_input_grad[4]:
            ctx.sort = _AsyncSort(feat, ex_idx, ex_base, item_idx, item_base, field_base)
        return loss if labels.dtype == torch.float64 else loss.to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        ex_idx, ex_base, item_idx, item_base, feat, bias_table = ctx.args
        gt = ctx.gt
        if not _UNIT_LOSS_GRAD:   # an upstream factor other than 1 (loss * c): scale the saved records
            gt = gt * g.to(torch.float32)
        fg = ctx.sort.join()
        dim = feat.shape[1]
        if ctx.loss_final is not None:
            ops._loss_final_pending, ctx.loss_final = ctx.loss_final, None      # taken by the fg_reduce call below
        gfeat, gbias = table_backward_fm(feat, bias_table, fg, gt[:, :dim], gt[:, dim], ex_idx.shape[1], None,
                                         premultiplied=True)
        if ops._loss_final_pending is not None:
            ops._loss_final_pending = None
            raise RuntimeError("FMLoss: the backward pass did not reach rx_fg_reduce; the deferred loss was not written")
        return (None, None, None, None, gfeat, gbias, None, None, None)


# ------------------------------------------------------------------ tower-input assembly
class GatherConcat(Function):
    """out[M, W] assembled from gathered rows (rx_gather_concat).  ``spec`` is a list of
    ``(param_slot, idx, col, div, stride, k, base[, bag])``; ``tables`` the distinct table parameters.
    A block with ``bag`` > 1 is the sum of the rows of ``bag`` consecutive index elements."""

    @staticmethod
    def forward(ctx, spec, out_rows, out_cols, *tables):
        spec = [tuple(s) + (1,) if len(s) == 7 else tuple(s) for s in spec]
        segs = [Segment(tables[slot], idx, col, div, stride, k, base, bag) for (slot, idx, col, div, stride, k, base, bag) in spec]
        ctx.spec, ctx.tables, ctx.out_rows = spec, tables, out_rows
        return ops.gather_concat(segs, out_rows, out_cols)

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        grads = []
        for slot, table in enumerate(ctx.tables):
            mine = [s for s in ctx.spec if s[0] == slot]
            dim = table.shape[1]
            n_each = [ctx.out_rows // s[3] for s in mine]
            pooled = any(s[7] > 1 for s in mine)
            occ = torch.empty(sum(n_each), dim, dtype=torch.float32, device=dout.device)
            idx_parts, occ_parts = [], []
            o = 0
            for s, n in zip(mine, n_each):
                _, idx, col, div, stride, k, base, bag = s
                ops.gather_concat_bwd(dout, col, dim, div, out=occ[o:o + n])
                if bag > 1:      # every member of a bag receives the bag's gradient
                    used = idx.reshape(-1, stride)[:n, k:k + bag].reshape(-1)
                    occ_parts.append(occ[o:o + n].repeat_interleave(bag, 0))
                else:
                    used = idx.reshape(-1, stride)[:n, k] if stride > 1 else idx.reshape(-1)[:n]
                    occ_parts.append(occ[o:o + n])
                o += n
                idx_parts.append(used + base if base else used)
            flat = idx_parts[0] if len(idx_parts) == 1 else torch.cat(idx_parts)
            if pooled:
                occ = occ_parts[0] if len(occ_parts) == 1 else torch.cat(occ_parts)
            si = sorted_index(flat, table.shape[0])
            grads.append(table_backward(table, si, occ))
        return (None, None, None, *grads)


# ------------------------------------------------------------------ GMF product feeding a GEMM
class GMFMul(Function):
    @staticmethod
    def forward(ctx, users, items, U, V):
        ctx.args = (users, items, U, V)
        return ops.gmf_mul_fwd(U, V, users, items)

    @staticmethod
    def backward(ctx, dout):
        users, items, U, V = ctx.args
        user_occ, item_occ = ops.gmf_mul_bwd(dout.contiguous(), U, V, users, items)
        gU, gV = tables_backward((U, V), (sorted_index(users, U.shape[0]), sorted_index(items, V.shape[0])), (user_occ, item_occ))
        return None, None, gU, gV


# ------------------------------------------------------------------ NeuMF: forward + loss + backward in one launch
class NeuMFLoss(Function):
    """List-wise NeuMF loss (embedding_dim 64, default tower) through rx_neumf_step: the forward launch already
    computes every gradient -- a pair's dL/dlogit is known as soon as its logit is -- so the backward only hands
    them on: the four tables' per-row gradients to the sort + segmented reduce (or the fused row update), the
    eight weight gradients as views of one flat buffer."""

    @staticmethod
    def forward(ctx, users, items, labels, p_drop, seed, ug, ig, um, im, w1, b1, w2, b2, wg, bg, wf, bf):
        out = ops.neumf_step((ug, ig, um, im), users, items, labels.contiguous(), (w1, b1, w2, b2, wg, bg, wf, bf),
                             p_drop=p_drop, seed=seed)
        loss, ctx.du_g, ctx.du_m, ctx.di_g, ctx.di_m, ctx.wgrad, _ = out
        ctx.idx = (users, items)
        ctx.tables = (ug, ig, um, im)
        ctx.shapes = [tuple(t.shape) for t in (w1, b1, w2, b2, wg, bg, wf, bf)]
        return loss if labels.dtype == torch.float64 else loss.to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        users, items = ctx.idx
        ug, ig, um, im = ctx.tables
        grads = [ctx.du_g, ctx.di_g, ctx.du_m, ctx.di_m, ctx.wgrad]
        if not _UNIT_LOSS_GRAD:       # an upstream factor other than 1 (loss * c)
            f = g.to(torch.float32)
            grads = [t * f for t in grads]
        du_g, di_g, du_m, di_m, wgrad = grads
        su = sorted_index(users, ug.shape[0])
        si = sorted_index(items, ig.shape[0])
        gug, gig, gum, gim = tables_backward((ug, ig, um, im), (su, si, su, si), (du_g, di_g, du_m, di_m))
        outs, o = [], 0
        for shp in ctx.shapes:
            n = 1
            for d in shp:
                n *= d
            outs.append(wgrad[o:o + n].reshape(shp))
            o += n
        return (None, None, None, None, None, gug, gig, gum, gim) + tuple(outs)


# ------------------------------------------------------------------ Linear (+ReLU) on the matrix cores
class LinearAct(Function):
    """Linear (+ ReLU (+ Dropout)).  ``drop_p`` > 0: the Dropout behind a Linear + ReLU.  Its forward is
    torch's own fused dropout (one vectorised pass, mask from torch's generator, safe under graph
    capture); its backward is folded into the ReLU's: the pair's output is positive exactly where the
    ReLU was open and the element kept, so one pass ``dz = y > 0 ? dy / (1 - p) : 0`` over the saved output
    replaces the masked scaling and the ReLU backward, and neither the mask nor the ReLU's own output
    is kept.  (Applying a keep mask in the product's epilogue was measured too: drawing the mask as a
    tensor costs more than the fused dropout pass it replaces.)"""

    @staticmethod
    def forward(ctx, x, w, b, act, bf16, drop_p=0.0, into=None):
        # ``into``: a one-element list holding the [M, N] destination (e.g. a column block of the buffer a
        # later layer reads: see JoinColumns), handed over outside autograd's view so that writing it is
        # not recorded as an in-place change of that buffer
        if drop_p and not act:
            raise ValueError("the fused dropout comes with the ReLU in front of it")
        if into is not None and (act or drop_p):
            raise ValueError("a destination block is for layers whose output the backward does not read")
        y = ops.linear_fwd(x, w, b, act, bf16, out=into[0] if into is not None else None)
        if into is not None:
            y = y.detach()
        if drop_p:
            y = torch.nn.functional.dropout(y, float(drop_p), True)
        ctx.save_for_backward(x, w, y if act else None)
        ctx.act, ctx.bf16, ctx.has_b = act, bf16, b is not None
        ctx.scale = 1.0 / (1.0 - float(drop_p)) if drop_p else 1.0
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        if ctx.act:
            dy = ops.relu_bwd(dy.contiguous(), y, ctx.scale)
        elif not (dy.dim() == 2 and dy.stride(1) == 1 and dy.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0):
            # (a column block of a wider gradient -- the backward of torch.cat -- is read in place:
            # the products take a row stride)
            dy = dy.contiguous()
        dx = ops.linear_bwd_x(dy, w, ctx.bf16) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2]):
            dw, db = ops.linear_bwd_w(dy, x, ctx.has_b, ctx.bf16)
        return dx, dw, db, None, None, None, None


def linear(x: Tensor, w: Tensor, b: Optional[Tensor], relu: bool = False, bf16: bool = False, drop_p: float = 0.0,
           into: Optional[Tensor] = None) -> Tensor:
    return LinearAct.apply(x, w, b, 1 if relu else 0, bf16, drop_p, [into] if into is not None else None)


class JoinColumns(Function):
    """``torch.cat(parts, dim=1)`` without the copy: the parts were WRITTEN as column blocks of one buffer
    by the layers that made them (``linear(..., into=block)``), so the forward only hands that buffer
    on, and the backward hands every part its column block of the gradient, in place.  ``holder``: a
    one-element list with the buffer; ``parts`` must tile its columns from left to right."""

    @staticmethod
    def forward(ctx, holder, *parts):
        buf = holder[0]
        col = 0
        for p in parts:
            if p.data_ptr() != buf.data_ptr() + 4 * col or p.shape[0] != buf.shape[0] or p.stride(0) != buf.stride(0):
                raise ValueError("JoinColumns: a part is not the next column block of the buffer")
            col += p.shape[1]
        if col != buf.shape[1]:
            raise ValueError("JoinColumns: the parts do not cover the buffer")
        ctx.widths = [p.shape[1] for p in parts]
        return buf.detach()

    @staticmethod
    def backward(ctx, d):
        out, col = [], 0
        for wdt in ctx.widths:
            out.append(d[:, col:col + wdt])
            col += wdt
        return (None,) + tuple(out)


# ------------------------------------------------------------------ DCN-v2 cross layer
class CrossLayer(Function):
    """x_{l+1} = x0 * (x_l W^T + b) + x_l."""

    @staticmethod
    def forward(ctx, x0, xl, w, b, bf16):
        u = ops.linear_fwd(xl, w, b, 0, bf16)
        xn = ops.cross_combine_fwd(x0, u, xl)
        ctx.save_for_backward(x0, xl, w, u)
        ctx.bf16 = bf16
        return xn

    @staticmethod
    def backward(ctx, dxn):
        x0, xl, w, u = ctx.saved_tensors
        if dxn.stride(-1) != 1:
            dxn = dxn.contiguous()
        dx0 = ops._padded(x0.shape[0], x0.shape[1], x0.device)
        dx0.zero_()
        du, dxl = ops.cross_combine_bwd(dxn, x0, u, dx0)
        ops.linear_bwd_x(du, w, ctx.bf16, out=dxl, accumulate=True)
        dw, db = ops.linear_bwd_w(du, xl, True, ctx.bf16)
        return dx0, dxl, dw, db, None


class CrossStack(Function):
    """A whole stack of cross layers, x_{l+1} = x0 * (x_l W_l^T + b_l) + x_l with x_0 = x0, as one
    autograd node: the backward walk keeps ONE dX0 buffer for all layers (no zero fills, no
    per-layer gradient sums by autograd) and lets the grad-X GEMM of an inner layer accumulate
    straight into the buffer the incoming gradient lives in."""

    @staticmethod
    def forward(ctx, x0, bf16, *wb):
        ws, bs = wb[0::2], wb[1::2]
        xs, us, x = [], [], x0
        for w, b in zip(ws, bs):
            u = ops.linear_fwd(x, w, b, 0, bf16)
            xs.append(x)
            us.append(u)
            x = ops.cross_combine_fwd(x0, u, x)
        ctx.save_for_backward(x0, *xs[1:], *us, *ws)
        ctx.bf16, ctx.L = bf16, len(ws)
        return x

    @staticmethod
    def backward(ctx, dxn):
        L = ctx.L
        saved = ctx.saved_tensors
        x0 = saved[0]
        xs = (x0,) + tuple(saved[1:L])
        us, ws = saved[L:2 * L], saved[2 * L:3 * L]
        d = dxn if dxn.stride(-1) == 1 else dxn.contiguous()
        dx0 = ops._padded(x0.shape[0], x0.shape[1], x0.device)
        grads = [None] * (2 * L)
        for l in range(L - 1, -1, -1):
            last, first = l == L - 1, l == 0
            # the incoming gradient of the last layer belongs to autograd: copy it (dxl) before the
            # GEMM adds to it; further down `d` is this function's own buffer
            du, dxl = ops.cross_combine_bwd_stack(d, x0, us[l], dx0, accumulate=not last, identity=first,
                                                  want_dxl=last and not first)
            target = dx0 if first else (dxl if last else d)
            ops.linear_bwd_x(du, ws[l], ctx.bf16, out=target, accumulate=True)
            grads[2 * l], grads[2 * l + 1] = ops.linear_bwd_w(du, xs[l], True, ctx.bf16)
            d = target
        return (dx0, None) + tuple(grads)


def cross_stack(x0: Tensor, layers, bf16: bool = False) -> Tensor:
    """``layers``: modules with ``weight`` [d, d] and ``bias`` [d] (nn.Linear), first layer first."""
    wb = []
    for m in layers:
        wb += [m.weight, m.bias]
    return CrossStack.apply(x0, bf16, *wb)


def h16_shapes_ok(x: Tensor, weights) -> bool:
    """Shapes the bf16-activation kernels take: fp32 GPU input with 16-byte addressable rows, every
    layer width a multiple of 4 (the last layer may have one output column, up to 4096 inputs)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 1 and x.shape[1] % 4 == 0
            and x.shape[1] >= 8 and x.stride(1) == 1 and x.stride(0) % 4 == 0):
        return False
    ws = list(weights)
    for i, w in enumerate(ws):
        n, k = w.shape
        if k % 4 != 0 or k < 8:
            return False
        if n % 4 != 0 and not (i == len(ws) - 1 and n == 1 and k <= 4096):
            return False
    return True


class CrossStack16(Function):
    """CrossStack with bf16 operand copies in HBM: every x_l is kept in fp32 for the residual chain
    and as the bf16 copy the matrix products read (written by the same combine kernel); du exists in
    bf16 only.  Same arithmetic as CrossStack(bf16=True) -- the products round these operands to
    bf16 there too -- except that the bias gradient sums the rounded du.  Returns (x_L, bf16 copy)."""

    @staticmethod
    def forward(ctx, x0, *wb):
        ws, bs = wb[0::2], wb[1::2]
        x16 = ops.to_bf16(x0)
        xs16, us, x = [x16], [], x0
        for w, b in zip(ws, bs):
            if b is not None and x.shape[0] > 1:
                u, x, xn16 = ops.cross16_layer_fwd(xs16[-1], w, b, x0, x)     # product + combine in one launch
            else:
                u = ops.linear16_fwd(xs16[-1], w, b, relu=False, out_bf16=False)
                x, xn16 = ops.cross16_fwd(x0, u, x)
            us.append(u)
            xs16.append(xn16)
        ctx.save_for_backward(x0, *xs16[:-1], *us, *ws)
        ctx.L = len(ws)
        ctx.mark_non_differentiable(xs16[-1])
        return x, xs16[-1]

    @staticmethod
    def backward(ctx, dxn, _d16):
        L = ctx.L
        saved = ctx.saved_tensors
        x0 = saved[0]
        xs16, us, ws = saved[1:1 + L], saved[1 + L:1 + 2 * L], saved[1 + 2 * L:1 + 3 * L]
        d = dxn if (dxn.stride(-1) == 1 and dxn.stride(0) % 4 == 0) else ops._padded(*dxn.shape, dxn.device).copy_(dxn)
        dx0 = ops._padded(x0.shape[0], x0.shape[1], x0.device)
        grads = [None] * (2 * L)
        for l in range(L - 1, -1, -1):
            last, first = l == L - 1, l == 0
            du16, dxl = ops.cross16_bwd(d, x0, us[l], dx0, accumulate=not last, identity=first, want_dxl=last and not first)
            target = dx0 if first else (dxl if last else d)
            ops.linear16_bwd_x(du16, ws[l], out=target, accumulate=True)
            grads[2 * l], grads[2 * l + 1] = ops.linear16_bwd_w(du16, xs16[l], True)
            d = target
        return (dx0,) + tuple(grads)


class MLPStack16(Function):
    """Linear + ReLU layers (the last one without ReLU) as one autograd node with bf16 activations in
    HBM: hidden outputs are stored in bf16 only, the backward walk keeps the pre-activation
    gradients in bf16 and applies each ReLU gate inside the epilogue of the product behind it.
    ``x16``: the input's bf16 copy when the producer already made one (else None)."""

    @staticmethod
    def forward(ctx, x, x16, *wb):
        ws, bs = wb[0::2], wb[1::2]
        acts = [x16 if x16 is not None else ops.to_bf16(x)]
        L = len(ws)
        for i, (w, b) in enumerate(zip(ws, bs)):
            last = i == L - 1
            acts.append(ops.linear16_fwd(acts[-1], w, b, relu=not last, out_bf16=not last))
        ctx.save_for_backward(*acts[:-1], *ws)
        ctx.L, ctx.has_b = L, [b is not None for b in bs]
        return acts[-1]

    @staticmethod
    def backward(ctx, dy):
        L = ctx.L
        saved = ctx.saved_tensors
        acts, ws = saved[:L], saved[L:2 * L]
        d16 = ops.to_bf16(dy.contiguous())
        grads = [None] * (2 * L)
        dx = None
        for l in range(L - 1, -1, -1):
            dw, db = ops.linear16_bwd_w(d16, acts[l], ctx.has_b[l])
            grads[2 * l], grads[2 * l + 1] = dw, db
            if l > 0:
                d16 = ops.linear16_bwd_x(d16, ws[l], gate16=acts[l], out_bf16=True)
            elif ctx.needs_input_grad[0]:
                dx = ops.linear16_bwd_x(d16, ws[l], out_bf16=False)
        return (dx, None) + tuple(grads)


class BNTower16(Function):
    """The reference's BatchNorm tower -- [Linear, BatchNorm1d, ReLU, Dropout] * n + Linear(., 1)
    (recsys/models/neuralcf/deep_factorization_machine.py:6-24) -- as one autograd node with bf16
    operand storage: every Linear reads a bf16 activation (the input's copy, then the bf16 output of
    the normalisation in front of it) and writes fp32 for the statistics pass; the backward walk
    keeps the gradient that enters a normalisation in fp32 and the one that leaves it -- an operand
    of two matrix products -- in bf16.  Same arithmetic as the per-layer nodes with ``bf16=True``.
    ``cfg``: per hidden block (momentum, eps, p, batch statistics?, update running buffers?);
    tensors per block: w, b, gamma, beta, running_mean, running_var; then the last layer's w, b."""

    @staticmethod
    def forward(ctx, x, cfg, *t):
        nb = len(cfg)
        a16 = [ops.to_bf16(x)]
        zs, stats, keeps = [], [], []
        for i, (momentum, eps, p, batch_stats, update) in enumerate(cfg):
            w, b, gamma, beta, rm, rv = t[6 * i:6 * i + 6]
            z = ops.linear16_fwd(a16[-1], w, b, relu=False, out_bf16=False)
            if batch_stats:
                if z.shape[0] < 2 and update:
                    raise ValueError("Expected more than 1 value per channel when training")
                mean, rstd = ops.bn_stats(z, eps, momentum if update else 0.0, rm if update else None, rv if update else None)
            else:
                mean, rstd = rm, torch.rsqrt(rv + eps)
            keep, keep_scale = None, 1.0
            if p > 0.0:
                keep = torch.empty(z.shape, dtype=torch.uint8, device=z.device).bernoulli_(1.0 - p)
                keep_scale = 1.0 / (1.0 - p)
            a16.append(ops.bn_apply16(z, mean, rstd, gamma, beta, True, keep, keep_scale))
            zs.append(z); stats += [mean, rstd]; keeps.append(keep_scale)
        w_last, b_last = t[6 * nb], t[6 * nb + 1]
        y = ops.linear16_fwd(a16[-1], w_last, b_last, relu=False, out_bf16=False)
        ctx.save_for_backward(*a16, *zs, *stats, *[t[6 * i + 2] for i in range(nb)], *[t[6 * i] for i in range(nb)], w_last)
        ctx.nb, ctx.keeps, ctx.cfg, ctx.has_b_last = nb, keeps, cfg, b_last is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        nb = ctx.nb
        s = ctx.saved_tensors
        a16, zs = s[:nb + 1], s[nb + 1:2 * nb + 1]
        stats = s[2 * nb + 1:4 * nb + 1]
        gammas, ws, w_last = s[4 * nb + 1:5 * nb + 1], s[5 * nb + 1:6 * nb + 1], s[6 * nb + 1]
        grads = [None] * (6 * nb + 2)
        d16 = ops.to_bf16(dy.contiguous())
        grads[6 * nb], db = ops.linear16_bwd_w(d16, a16[nb], ctx.has_b_last)
        grads[6 * nb + 1] = db
        dout = ops.linear16_bwd_x(d16, w_last, out_bf16=False)               # fp32: it enters a normalisation
        dx = None
        for i in range(nb - 1, -1, -1):
            batch_stats = ctx.cfg[i][3]
            dz16, dgamma, dbeta = ops.bn_bwd16(dout, zs[i], a16[i + 1], stats[2 * i], stats[2 * i + 1], gammas[i], True,
                                               ctx.keeps[i], batch_stats)
            grads[6 * i + 2], grads[6 * i + 3] = dgamma, dbeta
            grads[6 * i], grads[6 * i + 1] = ops.linear16_bwd_w(dz16, a16[i], True)
            if i > 0:
                dout = ops.linear16_bwd_x(dz16, ws[i], out_bf16=False)
            elif ctx.needs_input_grad[0]:
                dx = ops.linear16_bwd_x(dz16, ws[i], out_bf16=False)
        return (dx, None) + tuple(grads)


def bn_tower16(seq, x: Tensor):
    """Run ``seq`` through BNTower16 if it is [Linear, BatchNorm1d(affine), ReLU, Dropout] * n +
    [Linear(., 1)] with widths the bf16-activation kernels take; returns None otherwise."""
    import torch.nn as nn
    mods = list(seq)
    if len(mods) < 5 or (len(mods) - 1) % 4 != 0:
        return None
    blocks = [mods[i:i + 4] for i in range(0, len(mods) - 1, 4)]
    last = mods[-1]
    kinds = (nn.Linear, nn.BatchNorm1d, nn.ReLU, nn.Dropout)
    if not isinstance(last, nn.Linear) or last.out_features != 1 or last.bias is None:
        return None
    for blk in blocks:
        if not all(isinstance(m, k) for m, k in zip(blk, kinds)):
            return None
        if not blk[1].affine or blk[0].bias is None or blk[0].out_features % 4 != 0:
            return None
    if not h16_shapes_ok(x, [b[0].weight for b in blocks] + [last.weight]):
        return None
    cfg, tensors = [], []
    for lin, bn, _, drop in blocks:
        training = bn.training
        momentum = 0.0
        update = training and bn.track_running_stats
        if update:
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
            momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        batch_stats = training or not bn.track_running_stats
        cfg.append((momentum, bn.eps, float(drop.p) if drop.training else 0.0, batch_stats, update))
        tensors += [lin.weight, lin.bias, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                    bn.running_var if bn.track_running_stats else None]
    tensors += [last.weight, last.bias]
    return BNTower16.apply(x, tuple(cfg), *tensors)


# ------------------------------------------------------------------ loss
class BatchNormAct(Function):
    """BatchNorm1d (+ ReLU) (+ Dropout) of the reference's BN towers in three launches forward
    (statistics, their join, apply) and three backward.  Training mode normalises with the batch
    statistics and updates the running buffers like torch.nn.BatchNorm1d; evaluation mode uses
    the running buffers.  The dropout keep mask is drawn by torch (its generator, graph-safe)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu, p):
        batch_stats = training or running_mean is None
        if batch_stats:
            if x.shape[0] < 2 and training:
                raise ValueError("Expected more than 1 value per channel when training")
            mean, rstd = ops.bn_stats(x, eps, momentum if training else 0.0,
                                      running_mean if training else None, running_var if training else None)
        else:
            mean, rstd = running_mean, torch.rsqrt(running_var + eps)
        keep, keep_scale = None, 1.0
        if training and p > 0.0:
            keep = torch.empty(x.shape, dtype=torch.uint8, device=x.device).bernoulli_(1.0 - p)
            keep_scale = 1.0 / (1.0 - p)
        out = ops.bn_apply(x, mean, rstd, gamma, beta, relu, keep, keep_scale)
        ctx.gate = bool(relu or keep is not None)
        ctx.keep_scale, ctx.batch_stats = keep_scale, batch_stats
        ctx.save_for_backward(x, out if ctx.gate else None, mean, rstd, gamma)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, out, mean, rstd, gamma = ctx.saved_tensors
        if dout.stride(-1) != 1:
            dout = dout.contiguous()
        dx, dgamma, dbeta = ops.bn_bwd(dout, x, out, mean, rstd, gamma, ctx.gate, ctx.keep_scale, ctx.batch_stats)
        return dx, (dgamma if gamma is not None else None), (dbeta if gamma is not None else None), None, None, None, None, None, None, None


def batch_norm_act(x: Tensor, bn: "torch.nn.BatchNorm1d", relu: bool, p: float) -> Tensor:
    """``Dropout(p)(ReLU(bn(x)))`` (either may be absent) with torch.nn.BatchNorm1d's state handling."""
    training = bn.training
    momentum = 0.0
    if training and bn.track_running_stats:
        if bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return BatchNormAct.apply(x, bn.weight, bn.bias, rm, rv, training, momentum, bn.eps, relu, p if training else 0.0)


_UNIT_LOSS_GRAD = False
_ONES = {}


def backward_from_loss(loss: Tensor) -> None:
    """``loss.backward()`` for a loss that is the direct output of ``bce_with_logits``: the root
    gradient is a cached device 1 and the loss function hands on its saved dL/dlogit unscaled, which
    removes three tiny launches (fill, cast, multiply) from the step's critical path."""
    global _UNIT_LOSS_GRAD
    key = (loss.device, loss.dtype)
    one = _ONES.get(key)
    if one is None:
        one = _ONES[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
    if not (loss.grad_fn is not None and type(loss.grad_fn).__name__ in ("BCEWithLogitsBackward", "FMLossBackward", "MFLossBackward", "NeuMFLossBackward")):
        loss.backward(gradient=one)
    else:
        _UNIT_LOSS_GRAD = True
        try:
            loss.backward(gradient=one)
        finally:
            _UNIT_LOSS_GRAD = False
    if loss.is_cuda:
        join_branches(loss.device)


class BCEWithLogits(Function):
    """Mean BCE-with-logits; fp64 loss when the labels are fp64, like the reference."""

    @staticmethod
    def forward(ctx, logits, labels):
        loss, dl = ops.bce_with_logits(logits.contiguous(), labels.contiguous(), True)
        ctx.save_for_backward(dl)
        ctx.shape = logits.shape
        return loss if labels.dtype == torch.float64 else loss.to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        if _UNIT_LOSS_GRAD:  # set by backward_from_loss: g is exactly 1
            return dl.reshape(ctx.shape), None
        return (dl * g.to(torch.float32)).reshape(ctx.shape), None


def bce_with_logits(logits: Tensor, labels: Tensor) -> Tensor:
    return BCEWithLogits.apply(logits, labels)
