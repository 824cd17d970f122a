"""Fused optimizer for the device-resident train step.

The reference trains every model with a dense torch optimizer over all parameters after a
global ``clip_grad_norm_`` (recsys/utils.py:101-103, recsys/pipelines/*: ``Adam(lr=1e-3)``).
That still works unchanged with the drop-in modules.  ``FusedOptimizer`` is the MI355X-native
alternative named by the build's north star: embedding tables get a *sparse rule*, so the
backward kernel that reduces a row's gradient also updates that row (SGD / Adagrad /
row-wise Adagrad are exactly their dense torch counterparts on the touched rows and leave
untouched rows alone), and tower weights are stepped by one fused launch per tensor.

With ``clip_norm`` set, the update is two-phase, as a global norm requires: backward writes
compact per-row gradients plus sum-of-squares partials, ``step()`` forms the global scale on
the device (no host sync) and then applies the row and dense updates with it.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import os

import torch

from . import _lib, ops

_SEPARATE_CLIP_LAUNCH = os.environ.get("RX_SEPARATE_CLIP") == "1"   # A/B timing of the clip factor formed inside the row update
_ONE_LAUNCH_PER_TABLE = os.environ.get("RX_NO_MULTI_TABLE") == "1"   # A/B timing of the multi-table launches

_NO_SPECULATION = os.environ.get("RX_NO_SPECULATIVE_SGD") == "1"   # A/B: clipped SGD over field groups always in two phases
# The same one-pass step on the general sorted path (rx_segreduce_fm, records of the old rows only): correct and
# tested, but OFF unless asked for.  Measured on fm_wide_d128 (125 M rows x 128, B = 65536): 0.54 ms per step against
# 0.565 in two phases while the clip is idle -- first-touch writes of the records cost nearly what the second pass
# saves -- and 1.0 ms when it bites (rows put back, two-phase step from T and g again), which it does on every
# step of that workload at clip_norm = 1.
_SPECULATE_SORTED = os.environ.get("RX_SPECULATIVE_SGD_SORTED") == "1"
_NO_FUSED_TAIL = os.environ.get("RX_NO_FUSED_TAIL") == "1"      # clip factor, row updates and dense update as three launches (A/B)
_FUSED_TAIL_MAX_ROWS, _FUSED_TAIL_MAX_PARTS = 1 << 18, 4096

_MODES = {"sgd": _lib.RX_SEG_SGD, "adagrad": _lib.RX_SEG_ADAGRAD, "rowwise_adagrad": _lib.RX_SEG_ROWWISE_ADAGRAD}


class SparseRule:
    """Attached to an embedding parameter as ``param._rx_sparse``; see autograd.table_backward."""

    def __init__(self, kind: str, lr: float, eps: float, owner: "FusedOptimizer"):
        self.kind, self.mode, self.lr, self.eps, self.owner = kind, _MODES[kind], lr, eps, owner
        self.state: Optional[torch.Tensor] = None

    def _state_for(self, param):
        if self.kind == "sgd":
            return None
        if self.state is None:
            shape = param.shape if self.kind == "adagrad" else (param.shape[0],)
            self.state = torch.zeros(shape, dtype=torch.float32, device=param.device)
        return self.state

    def apply(self, param, occ, si, dim=None, occ_ld=None, param2=None, occ2=None):
        table = param.data
        state = self._state_for(param)
        rule2 = getattr(param2, "_rx_sparse", None) if param2 is not None else None
        if param2 is not None and rule2 is None:
            raise RuntimeError("companion table must carry a sparse rule when its main table does")
        table2 = param2.data if param2 is not None else None
        state2 = rule2._state_for(param2) if rule2 is not None else None
        if self.owner.clip_norm is None:
            ops.segreduce(occ, si, self.mode, table, state=state, lr=self.lr, eps=self.eps, dim=dim, occ_ld=occ_ld,
                          occ2=occ2, target2=table2, state2=state2)
            return
        # With a clip pending nothing can move before the global norm is known: the reduce itself is
        # left to step(), which takes the tables of equal width in one launch.
        d = dim if dim is not None else (occ.shape[1] if occ.dim() == 2 else 1)
        self.owner._deferred.append((self, param, occ, si, d, occ_ld, rule2, param2, occ2))
        if _ONE_LAUNCH_PER_TABLE:
            self.owner._reduce_deferred()


def _rule_apply_fm(self, feat, bias_table, si, tsum, g, n_fields_ex, deep=None, premultiplied=False):
    """N == 1 FM: reduce + update straight from T and g (rx_segreduce_fm)."""
    table, state = feat.data, self._state_for(feat)
    rule2 = getattr(bias_table, "_rx_sparse", None) if bias_table is not None else None
    if bias_table is not None and rule2 is None:
        raise RuntimeError("companion table must carry a sparse rule when its main table does")
    table2 = bias_table.data if bias_table is not None else None
    state2 = rule2._state_for(bias_table) if rule2 is not None else None
    if isinstance(si, ops.FieldGroups):
        # field-grouped occurrence list: compact gradients live at the last position of each run
        if self.owner.clip_norm is None:
            ops.fg_reduce(tsum, g, si, self.mode, table, state=state, target2=table2, state2=state2,
                          lr=self.lr, eps=self.eps, deep=deep, premultiplied=premultiplied)
            return
        dim = tsum.shape[1]
        sq_out = self.owner._sq_slice(si.n, dim, table.device, ops.fg_reduce_grid(si.n, dim))
        if self.owner._pending:
            # not the first backward pass since the last step (gradient accumulation): rows that an earlier
            # pass of this step already moved go back first, and this pass keeps to the two-phase form
            self.owner._restore_speculative()
        elif (self.kind == "sgd" and not _NO_SPECULATION and self.owner.speculative
              and (rule2 is None or (rule2.kind == "sgd" and rule2.lr == self.lr))):
            # Clipped SGD in one pass over the distinct rows when the clip does not bite: the reduce takes the
            # step with a factor of 1 and leaves undo records; step() redoes the rows only when the norm turns
            # out to exceed the bound (rx_fg_reduce_args.undo).  Bit-identical to the two-phase form either way.
            line = (bias_table is not None and table2.data_ptr() == table.data_ptr() + 4 * dim
                    and table.stride(0) == 2 * dim and table2.stride(0) == 2 * dim)
            undo = torch.empty(si.n, 2 * dim, dtype=torch.float32, device=table.device)
            undo2 = torch.empty(si.n, 2, dtype=torch.float32, device=table.device) if bias_table is not None and not line else None
            sq = ops.fg_reduce(tsum, g, si, _lib.RX_SEG_SGD_SPEC, table, target2=table2, lr=self.lr, want_sq=True, sq_out=sq_out,
                               deep=deep, premultiplied=premultiplied, undo=undo, undo2=undo2)
            self.owner._pending.append((self, feat, undo, si, sq, (rule2, bias_table, undo2) if bias_table is not None else None, "spec"))
            self.owner._has_spec = True
            return
        compact = torch.empty(si.n, dim, dtype=torch.float32, device=table.device)
        compact2 = torch.empty(si.n, dtype=torch.float32, device=table.device) if bias_table is not None else None
        sq = ops.fg_reduce(tsum, g, si, _lib.RX_SEG_COMPACT, compact, table=table, target2=compact2, want_sq=True,
                           sq_out=sq_out, deep=deep, premultiplied=premultiplied)
        self.owner._pending.append((self, feat, compact, si, sq, (rule2, bias_table, compact2) if bias_table is not None else None, False))
        return
    if self.owner.clip_norm is None:
        ops.segreduce_fm(tsum, g, n_fields_ex, si, self.mode, table, state=state, target2=table2, state2=state2,
                         lr=self.lr, eps=self.eps, deep=deep)
        return
    dim = tsum.shape[1]
    if self.owner._pending:
        self.owner._restore_speculative()          # gradient accumulation: see the field-grouped branch above
    elif (self.kind == "sgd" and _SPECULATE_SORTED and not _NO_SPECULATION and self.owner.speculative and table.is_cuda and table.is_contiguous()
          and (rule2 is None or (rule2.kind == "sgd" and rule2.lr == self.lr))):
        # the same one-pass clipped SGD step on the general sorted path (rows wider than the field groups take,
        # N > 1, B > 65536).  The records keep only the rows of before the step (RX_SEG_SGD_SPEC_OLD: a 512-byte
        # row's record of gradient + row is eight new lines per distinct row, and cost more than the second pass
        # saved): when the clip bites, step() puts the rows back and takes the two-phase step from T and g again.
        undo = torch.empty(max(si.n, 1), dim, dtype=torch.float32, device=table.device)
        undo2 = torch.empty(max(si.n, 1), dtype=torch.float32, device=table.device) if bias_table is not None else None
        sq = ops.segreduce_fm(tsum, g, n_fields_ex, si, _lib.RX_SEG_SGD_SPEC_OLD, table, state=undo, target2=table2, state2=undo2,
                              lr=self.lr, want_sq=True, sq_out=self.owner._sq_slice(si.n, dim, table.device), deep=deep)
        self.owner._pending.append((self, feat, undo, si, sq, (rule2, bias_table, undo2) if bias_table is not None else None, "spec_old"))
        self.owner._again[id(feat)] = (tsum, g, n_fields_ex, deep)
        self.owner._has_spec = True
        return
    compact = torch.empty(max(si.n, 1), dim, dtype=torch.float32, device=table.device)
    compact2 = torch.empty(max(si.n, 1), dtype=torch.float32, device=table.device) if bias_table is not None else None
    sq = ops.segreduce_fm(tsum, g, n_fields_ex, si, _lib.RX_SEG_COMPACT, compact, table=table, target2=compact2,
                          want_sq=True, sq_out=self.owner._sq_slice(si.n, dim, table.device), deep=deep)
    self.owner._pending.append((self, feat, compact, si, sq, (rule2, bias_table, compact2) if bias_table is not None else None, False))


SparseRule.apply_fm = _rule_apply_fm


def before_table_read(param) -> None:
    """Called by the forward passes that gather from ``param``: rows that a speculative clipped step has
    already moved (backward done, step() still to come) are put back first, so that whatever runs between
    backward and step -- another forward + backward for gradient accumulation, an evaluation -- sees the
    weights of before the step, as with torch.optim.SGD."""
    rule = getattr(param, "_rx_sparse", None)
    if rule is not None and rule.owner._has_spec:
        rule.owner._restore_speculative()


class FusedOptimizer:
    """``FusedOptimizer(model, kind, lr, clip_norm=None, dense_kind=None)``.

    ``kind`` in {"sgd", "adagrad", "rowwise_adagrad"} applies to the parameters the model lists
    in ``model.sparse_parameters()``; all other parameters are stepped densely with
    ``dense_kind`` (default: "sgd" for sgd, "adagrad" otherwise; "adam" allowed).
    ``speculative``: whether clipped SGD over field groups takes its step inside the backward's reduce and redoes
    rows only if the clip bites (default: only for models without dense parameters).
    ``kind="dense_adam"``: no sparse rule at all -- dense table gradients and Adam on every parameter,
    the reference pipelines' optimizer (``from_torch`` builds it from a ``torch.optim.Adam``).
    API mirrors torch.optim: ``zero_grad()``, ``step()``."""

    def __init__(self, model, kind: str = "sgd", lr: float = 0.01, eps: float = 1e-10,
                 clip_norm: Optional[float] = None, dense_kind: Optional[str] = None,
                 betas=(0.9, 0.999), adam_eps: float = 1e-8, speculative: Optional[bool] = None):
        if kind not in _MODES and kind != "dense_adam":
            raise ValueError(f"unknown sparse optimizer {kind!r}")
        self.kind, self.lr, self.eps, self.clip_norm = kind, lr, eps, clip_norm
        self.dense_kind = dense_kind or ("sgd" if kind == "sgd" else "adam" if kind == "dense_adam" else "adagrad")
        if self.dense_kind not in ("sgd", "adagrad", "adam"):
            raise ValueError(f"unknown dense optimizer {self.dense_kind!r}")
        if kind == "dense_adam" and self.dense_kind != "adam":
            raise ValueError("kind='dense_adam' steps every parameter with Adam")
        self.betas, self.adam_eps = betas, adam_eps
        # kind="dense_adam": what the reference's pipelines build -- torch.optim.Adam over ALL parameters behind
        # clip_grad_norm_ (recsys/pipelines/neuralcf/neumf.py:10, recsys/utils.py:101-103).  No parameter carries a
        # sparse rule: the tables' gradients come out of the backward pass as dense [rows, dim] arrays (sort +
        # segmented reduce, RX_SEG_DENSE) and every tensor, tables included, takes a fused Adam step with the device-side
        # step count -- so this step, too, records into a HIP graph and replays (utils.LoopGraph).
        self.sparse: List[torch.nn.Parameter] = [] if kind == "dense_adam" else list(model.sparse_parameters())
        ids = {id(p) for p in self.sparse}
        self.dense: List[torch.nn.Parameter] = [p for p in model.parameters() if id(p) not in ids]
        self._dense_state = {}
        self.rules = []
        for p in self.sparse:
            rule = SparseRule(kind, lr, eps, self)
            p._rx_sparse = rule
            self.rules.append(rule)
        # Optimizer state exists from the start: a buffer first made (and zero-filled) inside a step that is
        # being recorded into a graph would be zeroed again by every replay.
        for p, rule in zip(self.sparse, self.rules):
            rule._state_for(p)
        if self.dense_kind == "adagrad":
            for p in self.dense:
                self._dense_state_for(p)
        elif self.dense_kind == "adam":
            for p in self.dense:
                self._dense_state_for(p)
        # Clipped SGD taken speculatively inside the reduce (see _rule_apply_fm) pays when the clip is idle and costs a
        # second full pass over the rows when it bites.  Default: on for models that are embedding tables only (the FM:
        # its gradient norm sits below the reference's bound of 1 after the first steps), off when a tower's
        # gradients enter the norm (FieldDeepFM at the benchmark shape clips on every step: 1.41 ms per step with
        # speculation, 1.35 without).  ``speculative=True / False`` decides it outright.
        self.speculative = bool(speculative) if speculative is not None else not self.dense
        self._pending = []
        self._has_spec = False                        # a pending entry whose rows the reduce already moved (speculative clipped SGD)
        self._again = {}                              # id(table) -> (T, g, fields, tower gradient) of a "spec_old" entry: what its redo starts from
        self._zero = None
        self._one = None
        self._deferred = []                            # reduces queued by the backward while a clip is set
        self._sq_buf: Optional[torch.Tensor] = None   # norm partials of all table reduces of a step, back to back
        self._sq_used = 0
        self._step = 0
        # Adam's step count also lives on the device and is advanced on the stream, so that a recorded graph
        # of the step reads the count of the replay, not the one of the recording.  Made here, never lazily
        # inside a step: a counter created (and filled) while a graph is being recorded would be reset to the
        # recording's value by every replay.
        self._step_dev = None
        if self.dense_kind == "adam" and self.dense and self.dense[0].is_cuda:
            self._step_dev = torch.zeros(1, dtype=torch.int32, device=self.dense[0].device)
        self.last_total_norm: Optional[torch.Tensor] = None

    @classmethod
    def from_torch(cls, optimizer, model, clip_norm=None):
        """The fused equivalent of a plain ``torch.optim.Adam`` / ``torch.optim.SGD`` built over all of
        ``model.parameters()`` (one parameter group, no weight decay, no amsgrad / momentum, no step taken yet),
        or None when the optimizer is anything else.  utils.train_loop(graph=True) uses it so that a pipeline
        written like the reference's -- ``optim.Adam(model.parameters(), lr=1e-3)`` -- replays as a HIP graph."""
        groups = getattr(optimizer, "param_groups", None)
        if not groups or len(groups) != 1 or len(optimizer.state) != 0:
            return None
        g = groups[0]
        if {id(p) for p in g["params"]} != {id(p) for p in model.parameters()}:
            return None
        if g.get("weight_decay", 0) or g.get("maximize", False):
            return None
        if type(optimizer) is torch.optim.Adam and not g.get("amsgrad", False):
            return cls(model, kind="dense_adam", lr=float(g["lr"]), clip_norm=clip_norm, betas=tuple(g["betas"]),
                       adam_eps=float(g["eps"]))
        if type(optimizer) is torch.optim.SGD and not g.get("momentum", 0) and not g.get("nesterov", False) and not g.get("dampening", 0):
            return cls(model, kind="sgd", lr=float(g["lr"]), clip_norm=clip_norm)
        return None

    def detach(self):
        """Remove the sparse rules (parameters go back to producing dense gradients)."""
        for p in self.sparse:
            if hasattr(p, "_rx_sparse"):
                del p._rx_sparse

    def _restore_speculative(self):
        """Put back the rows of pending speculative entries (factor 0 applied to their undo records) and
        mark the entries: step() then redoes them with the final clip factor whatever its value."""
        if not self._has_spec:
            return
        for k, e in enumerate(self._pending):
            rule, param, undo, si, sq, comp, flag = e
            if flag == "spec_old":
                rule2, param2, undo2 = comp if comp is not None else (None, None, None)
                ops.row_update(undo, si, _lib.RX_SEG_RESTORE, param.data, compact2=undo2, table2=param2.data if param2 is not None else None)
                self._pending[k] = (rule, param, undo, si, sq, comp, "redo_old")
                continue
            if flag != "spec":
                continue
            if self._zero is None or self._zero.device != param.device:
                self._zero = torch.zeros(1, dtype=torch.float32, device=param.device)
            rule2, param2, undo2 = comp if comp is not None else (None, None, None)
            update = ops.fg_row_update if isinstance(si, ops.FieldGroups) else ops.row_update
            update(undo, si, _lib.RX_SEG_SGD_SPEC_REDO, param.data, lr=rule.lr, scale=self._zero, compact2=undo2,
                   table2=param2.data if param2 is not None else None)
            self._pending[k] = (rule, param, undo, si, sq, comp, "redo")
        self._has_spec = False

    def zero_grad(self, set_to_none: bool = True):
        for p in self.dense:
            p.grad = None
        self._restore_speculative()        # gradients dropped without a step: the rows go back
        self._pending.clear()
        self._again.clear()
        self._deferred.clear()
        self._sq_used = 0

    def _reduce_deferred(self):
        """Compact gradients + norm partials of every queued table: tables of equal width share a
        launch pair (rx_segreduce_multi; up to 8, a table at most once per launch)."""
        todo, self._deferred = self._deferred, []
        while todo:
            d0, batch, rest, seen = todo[0][4], [], [], set()
            for e in todo:
                if e[4] == d0 and id(e[1]) not in seen and len(batch) < _lib.RX_SEG_MULTI_MAX:
                    batch.append(e)
                    seen.add(id(e[1]))
                else:
                    rest.append(e)
            todo = rest
            items, made = [], []
            for (rule, param, occ, si, d, occ_ld, rule2, param2, occ2) in batch:
                dev = param.device
                if occ.is_cuda:      # possibly made on a branch stream: this stream reads it now
                    occ.record_stream(torch.cuda.current_stream(dev))
                    if occ2 is not None:
                        occ2.record_stream(torch.cuda.current_stream(dev))
                compact = torch.empty(max(si.n, 1), d, dtype=torch.float32, device=dev)
                compact2 = torch.empty(max(si.n, 1), dtype=torch.float32, device=dev) if param2 is not None else None
                sq = self._sq_slice(si.n, d, dev)
                if sq is None:
                    sq = torch.empty(ops.segreduce_grid(si.n, d), dtype=torch.float32, device=dev)
                items.append(ops.TableReduce(occ, si, compact, d, occ_ld=occ_ld, occ2=occ2, target2=compact2, sq=sq))
                made.append((rule, param, compact, si, sq, (rule2, param2, compact2) if param2 is not None else None, False))
            ops.segreduce_multi(items, _lib.RX_SEG_COMPACT)
            self._pending.extend(made)

    def _update_pending_rows(self, scale):
        """Row updates left pending by a backward pass that ran with a clip set: applied with the clip
        factor, or unscaled when the clip was switched off between backward and step.  Tables of one
        width and rule share a launch."""
        plain = []
        for (rule, param, compact, si, _, comp, spec) in self._pending:
            if isinstance(si, ops.FieldGroups):
                rule2, param2, compact2 = comp if comp is not None else (None, None, None)
                if spec == "spec" and scale is None:
                    continue        # the clip was switched off after the backward pass: the rows already hold the plain step
                mode = {"spec": _lib.RX_SEG_SGD_SPEC, "redo": _lib.RX_SEG_SGD_SPEC_REDO}.get(spec, rule.mode)
                if spec and scale is None:      # rows put back earlier, clip switched off since: redo them unscaled
                    if self._one is None or self._one.device != param.device:
                        self._one = torch.ones(1, dtype=torch.float32, device=param.device)
                ops.fg_row_update(compact, si, mode, param.data, state=rule._state_for(param), lr=rule.lr,
                                  eps=rule.eps, scale=scale if (scale is not None or not spec) else self._one, compact2=compact2,
                                  table2=param2.data if param2 is not None else None,
                                  state2=rule2._state_for(param2) if rule2 is not None else None)
            elif spec in ("spec_old", "redo_old"):
                if spec == "spec_old" and scale is None:
                    continue        # clip switched off after the backward pass: the rows already hold the plain step
                # the two-phase step from T and g again -- every launch of it returns at once while the factor is 1
                # ("spec_old": the rows hold the step already), or runs in full ("redo_old": the rows were put back)
                rule2, param2, undo2 = comp if comp is not None else (None, None, None)
                tsum, g, n_fields_ex, deep = self._again[id(param)]
                gated = spec == "spec_old"
                t2 = param2.data if param2 is not None else None
                if gated:
                    ops.row_update(compact, si, _lib.RX_SEG_RESTORE_IF_CLIPPED, param.data, scale=scale, compact2=undo2, table2=t2)
                grads = torch.empty(max(si.n, 1), compact.shape[1], dtype=torch.float32, device=param.device)
                grads2 = torch.empty(max(si.n, 1), dtype=torch.float32, device=param.device) if param2 is not None else None
                ops.segreduce_fm(tsum, g, n_fields_ex, si, _lib.RX_SEG_COMPACT_IF_CLIPPED if gated else _lib.RX_SEG_COMPACT, grads,
                                 table=param.data, target2=grads2, scale=scale if gated else None, deep=deep)
                ops.row_update(grads, si, _lib.RX_SEG_SGD_IF_CLIPPED if gated else _lib.RX_SEG_SGD, param.data, lr=rule.lr, scale=scale,
                               compact2=grads2, table2=t2)
            elif spec:
                if spec == "spec" and scale is None:
                    continue        # clip switched off after the backward pass: the rows already hold the plain step
                rule2, param2, compact2 = comp if comp is not None else (None, None, None)
                if scale is None:
                    if self._one is None or self._one.device != param.device:
                        self._one = torch.ones(1, dtype=torch.float32, device=param.device)
                ops.row_update(compact, si, _lib.RX_SEG_SGD_SPEC if spec == "spec" else _lib.RX_SEG_SGD_SPEC_REDO, param.data,
                               lr=rule.lr, scale=scale if scale is not None else self._one, compact2=compact2,
                               table2=param2.data if param2 is not None else None)
            else:
                plain.append((rule, param, compact, si, comp))
        while plain:
            r0, d0 = plain[0][0], plain[0][2].shape[1]
            key0 = (r0.mode, r0.lr, r0.eps, d0)
            batch, rest, seen = [], [], set()
            for e in plain:
                rule = e[0]
                if ((rule.mode, rule.lr, rule.eps, e[2].shape[1]) == key0 and id(e[1]) not in seen
                        and len(batch) < (1 if _ONE_LAUNCH_PER_TABLE else _lib.RX_SEG_MULTI_MAX)):
                    batch.append(e)
                    seen.add(id(e[1]))
                else:
                    rest.append(e)
            plain = rest
            items = []
            for (rule, param, compact, si, comp) in batch:
                rule2, param2, compact2 = comp if comp is not None else (None, None, None)
                items.append(ops.TableReduce(compact, si, param.data, d0, state=rule._state_for(param), occ2=compact2,
                                             target2=param2.data if param2 is not None else None,
                                             state2=rule2._state_for(param2) if rule2 is not None else None))
            ops.row_update_multi(items, r0.mode, lr=r0.lr, eps=r0.eps, scale=scale)

    def _fused_tail(self, parts, parts64) -> bool:
        """Clip factor, row updates and dense update in one launch (rx_step_tail) when the whole tail of the step is
        of the plain kind: compact gradients of sorted tables under one rule and width (the MF and NCF families),
        at most 24 contiguous dense tensors under SGD or Adagrad, every partial sum in ``parts`` / ``parts64``.
        Same bits as the three launches it replaces; RX_NO_FUSED_TAIL=1 keeps those."""
        if _NO_FUSED_TAIL or not self._pending or len(self._pending) > _lib.RX_SEG_MULTI_MAX or _ONE_LAUNCH_PER_TABLE:
            return False
        r0 = self._pending[0][0]
        d0 = self._pending[0][2].shape[1]
        seen = set()
        for (rule, param, compact, si, _, comp, spec) in self._pending:
            if spec or isinstance(si, ops.FieldGroups) or not param.is_cuda or id(param) in seen:
                return False
            if (rule.mode, rule.lr, rule.eps, compact.shape[1]) != (r0.mode, r0.lr, r0.eps, d0):
                return False
            seen.add(id(param))
        if self.dense_kind == "adam":
            return False
        # every block of the fused launch adds up all partial sums before it touches a row: a gain for the small
        # lists of the MF / NCF families (launch-bound tails), a loss once the lists are long -- 0.65 against 0.55 ms
        # per step on 1.7 M occurrences of 512-byte rows
        n_parts = (int(parts.numel()) if parts is not None else 0) + (int(parts64.numel()) if parts64 is not None else 0)
        if sum(int(item[2].shape[0]) for item in self._pending) > _FUSED_TAIL_MAX_ROWS or n_parts > _FUSED_TAIL_MAX_PARTS:
            return False
        live = [(p, p.grad) for p in self.dense if p.grad is not None]
        if len(live) > _lib.RX_MULTI_MAX or any(not (p.data.is_contiguous() and g.is_contiguous() and p.is_cuda) for p, g in live):
            return False
        items = []
        for (rule, param, compact, si, _, comp, spec) in self._pending:
            rule2, param2, compact2 = comp if comp is not None else (None, None, None)
            items.append(ops.TableReduce(compact, si, param.data, d0, state=rule._state_for(param), occ2=compact2,
                                         target2=param2.data if param2 is not None else None,
                                         state2=rule2._state_for(param2) if rule2 is not None else None))
        dmode = _lib.RX_SEG_SGD if self.dense_kind == "sgd" else _lib.RX_SEG_ADAGRAD
        states = [self._dense_state_for(p) for p, _ in live] if dmode == _lib.RX_SEG_ADAGRAD else None
        scale, self.last_total_norm = ops.step_tail(items, r0.mode, r0.lr, r0.eps, [p.data for p, _ in live], [g for _, g in live], states,
                                                    dmode, self.lr, self.eps, parts, parts64, float(self.clip_norm))
        self._pending.clear()
        self._again.clear()
        self._has_spec = False
        self._sq_used = 0
        return True

    def _sq_slice(self, n: int, dim: int, device, need: Optional[int] = None):
        """Room for one reduce's norm partials in the step's shared buffer (None when it is full: the
        reduce then allocates its own and the step adds it separately).  ``need``: number of partials
        when the reduce is not rx_segreduce."""
        if need is None:
            need = ops.segreduce_grid(n, dim)
        if self._sq_buf is None or self._sq_buf.device != device:
            if torch.cuda.is_current_stream_capturing():
                return None
            self._sq_buf = torch.empty(1 << 16, dtype=torch.float32, device=device)
        if self._sq_used + need > self._sq_buf.numel():
            return None
        out = self._sq_buf[self._sq_used:self._sq_used + need]
        self._sq_used += need
        return out

    def _dense_state_for(self, p):
        st = self._dense_state.get(id(p))
        if st is None:
            if self.dense_kind == "adagrad":
                st = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
            elif self.dense_kind == "adam":
                st = (torch.zeros_like(p.data, memory_format=torch.contiguous_format), torch.zeros_like(p.data, memory_format=torch.contiguous_format))
            else:
                return None
            self._dense_state[id(p)] = st
        return st

    def _dense_update(self, p, scale):
        g = p.grad
        if g is None:
            return
        g = g.contiguous()
        w = p.data
        if self.dense_kind == "sgd":
            ops.dense_sgd(w, g, self.lr, scale)
        elif self.dense_kind == "adagrad":
            ops.dense_adagrad(w, g, self._dense_state_for(p), self.lr, self.eps, scale)
        else:
            st = self._dense_state_for(p)
            ops.dense_adam(w, g, st[0], st[1], self.lr, self.betas[0], self.betas[1], self.adam_eps,
                           self._step_dev if self._step_dev is not None else self._step, scale)

    def _dense_step_all(self, scale):
        """All dense parameters in launches of up to 24 tensors (same bits as one launch per tensor)."""
        live = [(p, p.grad.contiguous()) for p in self.dense if p.grad is not None]
        if not live:
            return
        if any(not p.data.is_contiguous() for p, _ in live) or not live[0][0].is_cuda:
            for p, _ in live:
                self._dense_update(p, scale)
            return
        if self.dense_kind == "adam":
            step = self._step_dev if self._step_dev is not None else self._step
            for i in range(0, len(live), _lib.RX_MULTI_MAX):
                chunk = live[i:i + _lib.RX_MULTI_MAX]
                st = [self._dense_state_for(p) for p, _ in chunk]
                ops.multi_dense_adam([p.data for p, _ in chunk], [g for _, g in chunk], [x[0] for x in st], [x[1] for x in st],
                                     self.lr, self.betas[0], self.betas[1], self.adam_eps, step, scale)
            return
        mode = _lib.RX_SEG_SGD if self.dense_kind == "sgd" else _lib.RX_SEG_ADAGRAD
        for i in range(0, len(live), _lib.RX_MULTI_MAX):
            chunk = live[i:i + _lib.RX_MULTI_MAX]
            states = None
            if mode == _lib.RX_SEG_ADAGRAD:
                states = []
                for p, _ in chunk:
                    states.append(self._dense_state_for(p))
            ops.multi_dense_update([p.data for p, _ in chunk], [g for _, g in chunk], mode, self.lr, self.eps,
                                   scale=scale, states=states)

    @property
    def step_count(self) -> int:
        """Optimizer steps taken so far (graph replays included: read from the device counter when there is one)."""
        return int(self._step_dev.item()) if self._step_dev is not None else self._step

    def step(self):
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        if not capturing:          # a recording executes nothing: only the device counter follows the replays
            self._step += 1
        some = self.sparse[0] if self.sparse else (self.dense[0] if self.dense else None)
        if some is not None and some.is_cuda:
            from .autograd import join_branches
            join_branches(some.device)     # a model's second-stream branch may still be writing gradients
            cur = torch.cuda.current_stream(some.device)
            for p in self.dense:           # dense gradients may have been made on a branch stream: read here
                if p.grad is not None:
                    p.grad.record_stream(cur)
        if self.dense_kind == "adam" and self.dense:
            # the step count also lives on the device, advanced on the stream: a recorded graph of this
            # step then reads the count of the replay, not the one of the recording
            if self._step_dev is not None:
                self._step_dev.add_(1)
        scale = None
        if self._deferred:
            self._reduce_deferred()
        if self.clip_norm is not None:
            # Global norm.  Dense gradients: one launch leaves fp64 partials for up to 24 tensors;
            # they and the per-block partials of the last table reduce ride in the launch that forms
            # the clip factor.  Anything beyond that goes through a device double first.
            acc = None
            grads = [p.grad.contiguous().reshape(-1) for p in self.dense if p.grad is not None]
            parts = [e[4] for e in self._pending if e[4] is not None]
            if self._sq_buf is not None and self._sq_used:
                # partials written back to back into the shared buffer count as one array
                base = self._sq_buf.data_ptr()
                own = [q for q in parts if not (base <= q.data_ptr() < base + self._sq_buf.numel() * 4)]
                parts = own + [self._sq_buf[:self._sq_used]]
            parts64 = ops.multi_sumsq_partials(grads[:_lib.RX_MULTI_MAX]) if grads else None
            rest = grads[_lib.RX_MULTI_MAX:]
            if rest or len(parts) > 1:
                dev = (self.dense[0] if self.dense else self.sparse[0]).device
                acc = torch.zeros(1, dtype=torch.float64, device=dev)
                for g in rest:
                    ops.sumsq_accumulate(g, acc)
                for sq in parts[:-1]:
                    ops.sum_accumulate(sq, acc)
            e0 = self._pending[0] if len(self._pending) == 1 else None
            if (e0 is not None and isinstance(e0[3], ops.FieldGroups) and e0[3].n > 0 and acc is None and parts64 is None
                    and len(parts) == 1 and parts[0].is_contiguous() and not _SEPARATE_CLIP_LAUNCH):
                # one field-grouped table and nothing else (the FM): every block of the row update adds the
                # partial sums itself, which takes the one-block clip launch off the step's critical path
                rule, param, compact, si, _, comp, spec = e0
                rule2, param2, compact2 = comp if comp is not None else (None, None, None)
                scale, self.last_total_norm = ops.fg_row_update(
                    compact, si, {"spec": _lib.RX_SEG_SGD_SPEC, "redo": _lib.RX_SEG_SGD_SPEC_REDO}.get(spec, rule.mode), param.data, state=rule._state_for(param), lr=rule.lr, eps=rule.eps,
                    compact2=compact2, table2=param2.data if param2 is not None else None,
                    state2=rule2._state_for(param2) if rule2 is not None else None,
                    clip_parts=parts[0], max_norm=float(self.clip_norm))
                self._pending.clear()
                self._again.clear()
            elif (parts or parts64 is not None) and acc is None and self._fused_tail(parts[-1] if parts else None, parts64):
                return
            elif parts or parts64 is not None:
                scale, self.last_total_norm = ops.clip_scale_parts(parts[-1] if parts else None, acc,
                                                                   float(self.clip_norm), parts64)
            else:
                dev = (self.dense[0] if self.dense else self.sparse[0]).device
                scale, self.last_total_norm = ops.clip_scale(torch.zeros(1, dtype=torch.float64, device=dev),
                                                             float(self.clip_norm))
        self._update_pending_rows(scale)
        self._pending.clear()
        self._again.clear()
        self._has_spec = False
        self._sq_used = 0
        self._dense_step_all(scale)
