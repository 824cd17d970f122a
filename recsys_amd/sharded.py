"""Row-sharded embedding tables across the GPUs of one node (no counterpart in the reference,
which is single-device).

Row ``r`` of a table lives on rank ``r % G`` at local row ``r // G`` (round-robin keeps the hot
low ids of a Zipf field spread over all ranks).  The batch is data parallel.  One lookup is

    bucketise local indices by owner   rx_bucketize_by_owner      (stable, by one radix pass)
    all-to-all counts, then indices    torch.distributed (RCCL over xGMI on GPUs)
    owners gather their rows           rx_gather_concat
    all-to-all rows back, un-permute   rx_unpermute_rows

and the backward ships per-occurrence gradient rows to the owners the same way; each owner then
runs the deterministic sort -> segmented reduce -> row update on what it received (arrival order is
(source rank, position), so results do not depend on timing).  The collective itself is issued by
torch.distributed on the current stream; everything around it is a C-ABI kernel.

``backend`` is the module providing the kernels (``recsys_amd.ops``).  Tests on machines without a
GPU pass a CPU stand-in with the same function names to exercise the exchange logic under gloo.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from . import _lib


def _ops():
    from . import ops
    return ops


class ShardedTable:
    """fp32 table of ``n_rows`` global rows x ``dim``, this rank holding rows r with r % G == rank."""

    def __init__(self, n_rows: int, dim: int, device, group=None, backend=None, init_std: float = 1.0,
                 seed: int = 0, full_weight: Optional[torch.Tensor] = None):
        self.group = group
        self.G = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_rows, self.dim, self.device = n_rows, dim, torch.device(device)
        self.ops = backend if backend is not None else _ops()
        self.local_rows = (n_rows - self.rank + self.G - 1) // self.G
        if full_weight is not None:
            self.weight = full_weight[self.rank::self.G].to(self.device).contiguous().float()
        else:
            g = torch.Generator(device=self.device).manual_seed(seed * 1000 + self.rank)
            self.weight = torch.randn(self.local_rows, dim, device=self.device, generator=g)
            if init_std != 1.0:
                self.weight.mul_(init_std)      # in place: the table may take a good part of the HBM
        # run the collectives even with one rank (exercises the RCCL calls on a one-GPU machine)
        self.force_exchange = False
        self.grad: Optional[torch.Tensor] = None      # dense local gradient (when no rule is set)
        self.rule = None                               # (mode, lr, eps, state) for the fused row update
        # two-phase update for a global gradient-norm clip: the owner-side reduce leaves compact
        # per-row gradients + sum-of-squares partials here, apply_pending(scale) updates the rows
        self.clip = False
        self.pending = None

    def set_rule(self, kind: str, lr: float, eps: float = 1e-10):
        mode = {"sgd": _lib.RX_SEG_SGD, "adagrad": _lib.RX_SEG_ADAGRAD, "rowwise_adagrad": _lib.RX_SEG_ROWWISE_ADAGRAD}[kind]
        state = None
        if kind == "adagrad":
            state = torch.zeros_like(self.weight)
        elif kind == "rowwise_adagrad":
            state = torch.zeros(self.local_rows, device=self.device)
        self.rule = (mode, lr, eps, state)

    # ---- exchange plan for one index list -------------------------------------------------
    def plan(self, idx: torch.Tensor, n_valid: Optional[torch.Tensor] = None):
        """``n_valid``: device count of the leading entries of ``idx`` that are real (a padded list of
        distinct rows); the one host read below then also yields that count."""
        counts, send_local, perm = self.ops.bucketize_by_owner(idx, self.G, self.n_rows, n_valid)
        if self.G == 1 and not self.force_exchange:
            m = idx.numel() if n_valid is None else int(counts.sum().item())
            return dict(send=[m], recv=[m], recv_idx=send_local[:m], perm=perm[:m], n=m)
        recv_counts = torch.empty_like(counts)
        dist.all_to_all_single(recv_counts, counts, group=self.group)
        both = torch.stack([counts, recv_counts]).tolist()   # the one host sync per lookup: split sizes must be host integers
        send, recv = both[0], both[1]
        m = sum(send)
        recv_idx = torch.empty(sum(recv), dtype=torch.int64, device=idx.device)
        dist.all_to_all_single(recv_idx, send_local[:m].contiguous(), recv, send, group=self.group)
        return dict(send=send, recv=recv, recv_idx=recv_idx, perm=perm[:m], n=m)

    def pack_fixed(self, idx: torch.Tensor, n_valid: Optional[torch.Tensor], capacity: int, out=None):
        """The local half of plan_fixed (no collective): owner buckets packed into ``capacity`` slots
        each.  Returns (packed local row ids [G * capacity], slot of every list entry [n], entries
        per owner [G])."""
        counts, send_local, perm = self.ops.bucketize_by_owner(idx, self.G, self.n_rows, n_valid)
        if out is not None:         # an earlier result, overwritten in place (device backend only)
            self.ops.pack_by_owner(send_local, perm, counts, capacity, max(self.n_rows // self.G, 1), out=(out[0], out[1]))
            out[2].copy_(counts)
            return out
        return self.ops.pack_by_owner(send_local, perm, counts, capacity, max(self.n_rows // self.G, 1)) + (counts,)

    def plan_fixed(self, idx: torch.Tensor, n_valid: Optional[torch.Tensor], capacity: int, packed=None):
        """Exchange plan with ``capacity`` slots per owner and equal all-to-all splits: no host-side
        sizes, so the step never waits for the host.  A group that does not fit sets the deferred
        RX_ERR_CAPACITY flag (``ops.raise_if_index_error``)."""
        packed, src_of, counts = packed if packed is not None else self.pack_fixed(idx, n_valid, capacity)
        recv_idx = packed
        if not (self.G == 1 and not self.force_exchange):
            recv_idx = torch.empty_like(packed)
            dist.all_to_all_single(recv_idx, packed, group=self.group)
        return dict(fixed=True, recv_idx=recv_idx, src_of=src_of, counts=counts, n_valid=n_valid, capacity=capacity, n=idx.numel())

    def sort_received_ahead(self, plan, side) -> None:
        """The owner-side sort of a fixed-capacity plan needs only the slot ids that arrived in the
        forward pass: issue it on ``side`` right after the index exchange, so that the backward pass
        finds it done instead of running it between the gradient exchange and the reduce."""
        recv_idx = plan["recv_idx"]
        if not recv_idx.is_cuda or recv_idx.numel() == 0:
            return
        side.wait_stream(torch.cuda.current_stream(recv_idx.device))      # the index exchange
        recv_idx.record_stream(side)
        with torch.cuda.stream(side):
            si = self.ops.sort_dedup(recv_idx, self.local_rows)
            done = torch.cuda.Event()
            done.record(side)
        plan["owner_sort"] = (si, done)

    def _swap(self, buf: torch.Tensor) -> torch.Tensor:
        """All-to-all of a [G * capacity, ...] buffer with equal splits."""
        if self.G == 1 and not self.force_exchange:
            return buf
        out = torch.empty_like(buf)
        dist.all_to_all_single(out, buf, group=self.group)
        return out

    def _to_owners(self, rows_in_send_order: torch.Tensor, plan) -> torch.Tensor:
        if self.G == 1 and not self.force_exchange:
            return rows_in_send_order
        out = torch.empty((sum(plan["recv"]),) + tuple(rows_in_send_order.shape[1:]), dtype=rows_in_send_order.dtype,
                          device=rows_in_send_order.device)
        dist.all_to_all_single(out, rows_in_send_order, plan["recv"], plan["send"], group=self.group)
        return out

    def _from_owners(self, rows_at_owner: torch.Tensor, plan) -> torch.Tensor:
        if self.G == 1 and not self.force_exchange:
            return rows_at_owner
        out = torch.empty((plan["n"],) + tuple(rows_at_owner.shape[1:]), dtype=rows_at_owner.dtype,
                          device=rows_at_owner.device)
        dist.all_to_all_single(out, rows_at_owner, plan["send"], plan["recv"], group=self.group)
        return out

    def gather(self, plan) -> torch.Tensor:
        """Rows for the planned index list, in the list's own order: [n, dim]."""
        seg = self.ops.Segment(self.weight, plan["recv_idx"], 0)
        mine = self.ops.gather_concat([seg], plan["recv_idx"].numel(), self.dim)
        if plan.get("fixed"):
            return self.ops.permute_rows(self._swap(mine), plan["src_of"], plan["n_valid"])
        back = self._from_owners(mine, plan)
        return self.ops.unpermute_rows(back, plan["perm"])

    def gather_packed(self, plan) -> torch.Tensor:
        """Fixed-capacity plans only: the requested rows as they arrive, [G * capacity, dim], entry j
        of the list sitting at row ``plan["src_of"][j]``.  Consumers that address rows through an
        index anyway compose the two maps instead of paying for a reordering copy."""
        seg = self.ops.Segment(self.weight, plan["recv_idx"], 0)
        return self._swap(self.ops.gather_concat([seg], plan["recv_idx"].numel(), self.dim))

    def reduce_packed(self, sent: torch.Tensor, plan, companion: Optional["ShardedTable"] = None,
                      sent2: Optional[torch.Tensor] = None):
        """Counterpart of gather_packed: ``sent`` [G * capacity, dim] holds one gradient row per slot
        (zeros in unused slots), ``sent2`` [G * capacity] the companion table's."""
        got = self._swap(sent)
        got2 = self._swap(sent2.reshape(-1, 1)).reshape(-1) if companion is not None else None
        self._owner_reduce(got, got2, plan, companion)

    def scatter_grad(self, occ: torch.Tensor, plan, companion: Optional["ShardedTable"] = None,
                     occ2: Optional[torch.Tensor] = None):
        """Send per-occurrence gradient rows (list order) to the owners and reduce / update there.
        ``companion`` / ``occ2`` [n]: a second table addressed by the same rows (the [rows, 1] bias
        table) and its per-occurrence gradients; they share this list's plan, sort and reduce."""
        got2 = None
        if plan.get("fixed"):
            slots = plan["recv_idx"].numel()
            sent = torch.zeros(slots, self.dim, dtype=torch.float32, device=occ.device)   # padding slots carry zero gradients
            got = self._swap(self.ops.unpermute_rows(occ.contiguous(), plan["src_of"], plan["n_valid"], out=sent))
            if companion is not None:
                sent2 = torch.zeros(slots, 1, dtype=torch.float32, device=occ.device)
                got2 = self._swap(self.ops.unpermute_rows(occ2.reshape(-1, 1).contiguous(), plan["src_of"],
                                                          plan["n_valid"], out=sent2)).reshape(-1)
        else:
            sent = self.ops.permute_rows(occ.contiguous(), plan["perm"])
            got = self._to_owners(sent, plan)
            if companion is not None:
                sent2 = self.ops.permute_rows(occ2.reshape(-1, 1).contiguous(), plan["perm"])
                got2 = self._to_owners(sent2, plan).reshape(-1)
        self._owner_reduce(got, got2, plan, companion)

    def _owner_reduce(self, got, got2, plan, companion):
        """What arrived from all ranks, one gradient row per received index: sort by local row, reduce
        the duplicates in that fixed order, then accumulate (no rule) or update in place."""
        if got.shape[0] == 0:
            return
        ahead = plan.get("owner_sort")
        if ahead is not None:       # sorted on a side stream while the forward pass ran (sort_received_ahead)
            si, done = ahead
            main = torch.cuda.current_stream(got.device)
            main.wait_event(done)
            for t in (si.sorted_pos, si.sorted_slot, si.uniq_rows, si.seg_off, si.n_uniq):
                if t is not None:
                    t.record_stream(main)
        else:
            si = self.ops.sort_dedup(plan["recv_idx"], self.local_rows)
        if self.rule is None:
            g = torch.zeros_like(self.weight)
            g2 = torch.zeros_like(companion.weight) if companion is not None else None
            self.ops.segreduce(got, si, _lib.RX_SEG_DENSE, g, dim=self.dim, occ2=got2, target2=g2)
            self.grad = g if self.grad is None else self.grad + g
            if companion is not None:
                companion.grad = g2 if companion.grad is None else companion.grad + g2
        else:
            mode, lr, eps, state = self.rule
            w2 = st2 = None
            if companion is not None:
                if companion.rule is None or companion.rule[0] != mode:
                    raise RuntimeError("a companion table must carry the same update rule as its main table")
                w2, st2 = companion.weight, companion.rule[3]
            if self.clip:
                compact = torch.empty(si.n, self.dim, dtype=torch.float32, device=got.device)
                compact2 = torch.empty(si.n, dtype=torch.float32, device=got.device) if companion is not None else None
                sq = self.ops.segreduce(got, si, _lib.RX_SEG_COMPACT, compact, dim=self.dim, occ2=got2, target2=compact2,
                                        want_sq=True)
                self.pending = (compact, compact2, si, sq, companion)
                return
            self.ops.segreduce(got, si, mode, self.weight, state=state, lr=lr, eps=eps, dim=self.dim,
                               occ2=got2, target2=w2, state2=st2)

    def apply_pending(self, scale: Optional[torch.Tensor]):
        """Second phase of a clipped step: rows (and the companion's entries) move by the compact
        gradients kept by the owner-side reduce, scaled by the device scalar ``scale``."""
        if self.pending is None:
            return
        compact, compact2, si, _, companion = self.pending
        self.pending = None
        mode, lr, eps, state = self.rule
        kw = {}
        if companion is not None:
            kw = dict(compact2=compact2, table2=companion.weight, state2=companion.rule[3])
        self.ops.row_update(compact, si, mode, self.weight, state=state, lr=lr, eps=eps, scale=scale, **kw)


class _ShardedLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, table, idx):
        plan = table.plan(idx)
        ctx.table, ctx.plan = table, plan
        return table.gather(plan)

    @staticmethod
    def backward(ctx, g):
        ctx.table.scatter_grad(g, ctx.plan)
        return None, None, None


def sharded_lookup(table: ShardedTable, idx: torch.Tensor, anchor: torch.Tensor) -> torch.Tensor:
    """Differentiable lookup.  ``anchor`` is any tensor that requires grad (keeps the node in the
    autograd graph; the table itself is updated by its owner inside backward)."""
    return _ShardedLookup.apply(anchor, table, idx.reshape(-1).contiguous())


class _ShardedLookupPair(torch.autograd.Function):
    """Rows of a table and of its companion for one index list: one plan, one sort, one reduce."""

    @staticmethod
    def forward(ctx, anchor, table, companion, idx):
        plan = table.plan(idx)
        ctx.table, ctx.companion, ctx.plan = table, companion, plan
        return table.gather(plan), companion.gather(plan)

    @staticmethod
    def backward(ctx, g, g2):
        ctx.table.scatter_grad(g, ctx.plan, ctx.companion, g2)
        return None, None, None, None


def sharded_lookup_pair(table: ShardedTable, companion: ShardedTable, idx: torch.Tensor, anchor: torch.Tensor):
    if companion.n_rows != table.n_rows or companion.G != table.G:
        raise ValueError("companion table must be sharded like its main table")
    return _ShardedLookupPair.apply(anchor, table, companion, idx.reshape(-1).contiguous())


class ShardedFieldFM(torch.nn.Module):
    """26-field FM (see models.fields.FieldFM) over row-sharded feature and bias tables.  Each rank
    feeds its own slice of the global batch; the FM interaction runs on the distinct rows it
    received, with the same rx_fm2 / rx_segreduce_fm kernels (the received rows act as a
    batch-local table addressed through the batch's inverse map)."""

    def __init__(self, field_dims, embedding_dim, device, group=None, backend=None, full_state=None):
        super().__init__()
        self.field_dims = [int(x) for x in field_dims]
        self.n_fields, self.dim = len(self.field_dims), embedding_dim
        self.ops = backend if backend is not None else _ops()
        n = sum(self.field_dims)
        offs = [0]
        for d in self.field_dims[:-1]:
            offs.append(offs[-1] + d)
        self.register_buffer("_offsets", torch.tensor(offs, dtype=torch.int64, device=device), persistent=False)
        fw = full_state["feature_embedding.weight"] if full_state else None
        bw = full_state["bias_embedding.weight"] if full_state else None
        self.feature = ShardedTable(n, embedding_dim, device, group, backend, seed=1, full_weight=fw)
        self.bias = ShardedTable(n, 1, device, group, backend, seed=2, full_weight=bw)
        self._anchor = torch.nn.Parameter(torch.zeros(1, device=device))
        self._pos = {}
        self._side_work = None      # one-shot callable run on the owner-side stream of the next training forward
        # slots per owner of the fixed-capacity exchange; None = exact sizes (one host read per step)
        self.capacity: Optional[int] = None

    def _flat(self, sparse):
        F = self.n_fields - 1
        glob = sparse + self._offsets[None, :]
        # occurrence order of the FM kernels: all (b, f < F) first, then the last field's B entries
        return torch.cat([glob[:, :F].reshape(-1), glob[:, F]]).contiguous()

    def calibrate(self, batches, slack: float = 1.25) -> int:
        """Pick the exchange capacity from sample batches: the largest number of distinct rows any
        rank sends to one owner (maximum over ranks), times ``slack``, rounded up to 256.  Host work,
        done once before training."""
        G = self.feature.G
        worst = 1
        for b in batches:
            uniq = torch.unique(self._flat(b["sparse"]))
            worst = max(worst, int(torch.bincount(uniq % G, minlength=G).max()))
        if dist.is_initialized() and G > 1:
            t = torch.tensor([worst], dtype=torch.int64, device=self._offsets.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.feature.group)
            worst = int(t.item())
        self.capacity = (int(worst * slack) + 255) // 256 * 256
        return self.capacity

    def prepare(self, inputs):
        """Index work of a later step that needs neither the tables nor a collective (occurrence
        list, sort + dedup, owner buckets packed into slots), issued on a side stream: call it for
        batch i+1 *before* launching step i and it overlaps that step.  Returns the batch dict with
        the result attached; ``forward`` picks it up and joins the side stream.  Without a
        fixed capacity (or on the CPU stand-in backend) the batch is returned as it is."""
        sparse = inputs["sparse"]
        if not self.capacity or not sparse.is_cuda or "_rx_shard_prep" in inputs:
            return inputs
        dev = sparse.device
        main = torch.cuda.current_stream(dev)
        side = self._pos.get("side")
        if side is None:
            side = self._pos["side"] = torch.cuda.Stream(device=dev)
        side.wait_stream(main)      # = the steps launched so far; the step launched next runs beside this
        with torch.cuda.stream(side):
            si, packed, flat, slot_of, occ_slot = self._index_work(sparse)
            done = torch.cuda.Event()
            done.record(side)
        out = dict(inputs)
        out["_rx_shard_prep"] = (si, packed, done, flat, slot_of, occ_slot)
        return out

    def _index_work(self, sparse, old=None):
        """The index work of prepare() on the current stream.  ``old``: an earlier result for a batch
        of the same shape, overwritten in place (buffers that keep their addresses)."""
        ops = self.ops
        if old is None:
            flat = ops.fm_flat_rows(sparse.contiguous(), self._offsets)
            si = ops.sort_dedup(flat, self.feature.n_rows, want_inverse=True)
            packed = self.feature.pack_fixed(si.uniq_rows, si.n_uniq, self.capacity)
            slot_of, occ_slot = ops.compose_slots(packed[1], si.inverse)      # occurrence -> distinct row -> slot
            return si, packed, flat, slot_of, occ_slot
        si, packed, _, flat, slot_of, occ_slot = old
        if flat.numel() != sparse.numel():
            raise ValueError("prepare_static: `out` was prepared for a batch of another shape")
        ops.fm_flat_rows(sparse.contiguous(), self._offsets, out=flat)
        ops.sort_dedup(flat, self.feature.n_rows, want_inverse=True, out=si)
        self.feature.pack_fixed(si.uniq_rows, si.n_uniq, self.capacity, out=packed)
        ops.compose_slots(packed[1], si.inverse, out=(slot_of, occ_slot))
        return si, packed, flat, slot_of, occ_slot

    def prepare_static(self, inputs, out=None):
        """prepare() for recorded steps (graph.ShardedGraphStep): the same index work on the *current*
        stream, no event, and with ``out`` (a batch this method prepared before) written into that
        batch's buffers in place, so that a graph recorded against them finds every later result at the
        same addresses.  Only the preparation is rewritten: the other entries of ``out`` (its index and
        label tensors) stay as they are, so ``out`` is meant for a static batch whose tensors ``inputs``
        are (or were copied into).  The caller orders producer and consumer (fork / join inside the
        recording)."""
        sparse = inputs["sparse"]
        if not self.capacity or not sparse.is_cuda:
            return inputs
        if out is not None:
            self._index_work(sparse, out["_rx_shard_prep"])
            return out
        si, packed, flat, slot_of, occ_slot = self._index_work(sparse)
        res = dict(inputs)
        res["_rx_shard_prep"] = (si, packed, None, flat, slot_of, occ_slot)
        return res

    @staticmethod
    def _join_prep(prep, device):
        """Make the current stream wait for a prepare() result and own its tensors from here on."""
        si, packed, done, flat, slot_of, occ_slot = prep
        if done is None:            # prepare_static: ordered by the caller, buffers never return to the allocator
            return si, packed, slot_of, occ_slot
        main = torch.cuda.current_stream(device)
        main.wait_event(done)
        for t in (si.sorted_pos, si.sorted_slot, si.uniq_rows, si.seg_off, si.n_uniq, si.inverse, packed[0], packed[1], packed[2], flat,
                  slot_of, occ_slot):
            t.record_stream(main)
        return si, packed, slot_of, occ_slot

    def set_rule(self, kind, lr, eps=1e-10, clip_norm: Optional[float] = None):
        """``clip_norm``: clip the global gradient norm (all tables on all ranks, plus any dense
        gradients handed to ``finish_step``) as torch.nn.utils.clip_grad_norm_ does in
        recsys/utils.py:101; the row updates then wait for ``finish_step``."""
        self.feature.set_rule(kind, lr, eps)
        self.bias.set_rule(kind, lr, eps)
        self.clip_norm = clip_norm
        self.feature.clip = clip_norm is not None
        self.last_total_norm = None

    def finish_step(self, dense_grads=()):
        """After backward.  With a clip norm: squared norms of the owner-side table gradients are
        summed over the ranks (one 8-byte all-reduce; the dense gradients, equal on every rank after
        their own all-reduce, are counted once), the clip factor is formed on the device, and the
        pending row updates run with it.  Returns the factor (device scalar) or None."""
        if getattr(self, "clip_norm", None) is None:
            return None
        acc = torch.zeros(1, dtype=torch.float64, device=self._offsets.device)
        if self.feature.pending is not None:
            self.ops.sum_accumulate(self.feature.pending[3], acc)
        if dist.is_initialized() and (self.feature.G > 1 or self.feature.force_exchange):
            dist.all_reduce(acc, group=self.feature.group)
        flat = [g.contiguous().reshape(-1) for g in dense_grads]
        if flat:
            scale, self.last_total_norm = self.ops.clip_scale_parts(None, acc, float(self.clip_norm),
                                                                    self.ops.multi_sumsq_partials(flat))
        else:
            scale, self.last_total_norm = self.ops.clip_scale(acc, float(self.clip_norm))
        self.feature.apply_pending(scale)
        return scale

    def _zero_base(self, F, device):
        key = ("base", F, str(device))
        if key not in self._pos:
            self._pos[key] = torch.zeros(F, dtype=torch.int64, device=device)
        return self._pos[key]

    def _iota(self, n, device):
        key = ("iota", str(device))
        t = self._pos.get(key)
        if t is None or t.numel() < n:
            t = self._pos[key] = torch.arange(n, dtype=torch.int64, device=device)
        return t[:n]

    def forward(self, inputs):
        sparse = inputs["sparse"]
        prep = inputs.get("_rx_shard_prep")
        flat = prep[3] if prep is not None else self._flat(sparse)
        return _ShardedFM.apply(self._anchor, self, flat, sparse.shape[0], self.n_fields - 1, None, None, prep)[0]

    def loss(self, inputs):
        return self.ops.bce_loss(self(inputs), inputs["label"])


class ShardedFieldDeepFM(ShardedFieldFM):
    """ShardedFieldFM plus the BatchNorm tower of models.fields.FieldDeepFM over [gathered rows,
    dense features].  The tables are row-sharded (model parallel); the tower is replicated (data
    parallel): every rank normalises with the statistics of its own slice of the batch, as
    DistributedDataParallel does, and ``allreduce_dense_grads`` sums the tower gradients over the
    ranks in one flat buffer before ``dense_sgd_step``.  The gradient of the tower input joins the
    rows' FM gradients inside rx_segreduce_fm, so no per-occurrence rows exist on this path either."""

    def __init__(self, field_dims, embedding_dim, device, n_dense=13, dropout=0.1, bf16=False, group=None,
                 backend=None, full_state=None):
        super().__init__(field_dims, embedding_dim, device, group, backend, full_state)
        from .models.fields import _bn_stack, _tower_pad
        self.n_dense, self.bf16 = n_dense, bf16
        d_in = self.n_fields * embedding_dim + n_dense
        self.pad = _tower_pad(d_in, n_dense)
        d_in += self.pad
        self.deep_cols = d_in
        self.deep_mlp = _bn_stack(d_in, [d_in, d_in // 2, d_in // 4], dropout).to(device)
        if full_state:
            own = {k[len("deep_mlp."):]: v for k, v in full_state.items() if k.startswith("deep_mlp.")}
            self.deep_mlp.load_state_dict(own)
        elif dist.is_initialized() and self.feature.G > 1:
            for t in list(self.deep_mlp.parameters()) + list(self.deep_mlp.buffers()):
                dist.broadcast(t.data, 0, group=group)   # replicas start equal
        self._flat_grad = None

    def dense_parameters(self):
        return list(self.deep_mlp.parameters())

    def forward(self, inputs):
        from .models.fields import _dense_input
        sparse = inputs["sparse"]
        B = sparse.shape[0]
        prep = inputs.get("_rx_shard_prep")
        flat = prep[3] if prep is not None else self._flat(sparse)
        logits, x = _ShardedFM.apply(self._anchor, self, flat, B, self.n_fields - 1, self.deep_cols,
                                     _dense_input(inputs, self.n_dense, B, self.pad), prep)
        return logits + self.ops.run_tower(self.deep_mlp, x, bf16=self.bf16)

    def allreduce_dense_grads(self):
        """Sum of the tower gradients over the ranks, one collective on a flat buffer (73 MB at the
        1 B-row x 128 shape; a ring is fine at that size)."""
        params = [p for p in self.dense_parameters() if p.grad is not None]
        if not params or not dist.is_initialized() or (self.feature.G == 1 and not self.feature.force_exchange):
            return
        n = sum(p.numel() for p in params)
        if self._flat_grad is None or self._flat_grad.numel() != n:
            self._flat_grad = torch.empty(n, dtype=torch.float32, device=params[0].device)
        torch._foreach_copy_(list(self._flat_grad.split([p.numel() for p in params])), [p.grad.reshape(-1) for p in params])
        dist.all_reduce(self._flat_grad, group=self.feature.group)
        torch._foreach_copy_([p.grad.reshape(-1) for p in params], list(self._flat_grad.split([p.numel() for p in params])))

    def dense_grads(self):
        return [p.grad for p in self.dense_parameters() if p.grad is not None]

    def dense_sgd_step(self, lr: float, scale: Optional[torch.Tensor] = None):
        """``scale``: the clip factor of ``finish_step`` (device scalar)."""
        params = [p for p in self.dense_parameters() if p.grad is not None]
        grads = [p.grad for p in params]
        with torch.no_grad():
            if scale is not None:
                grads = torch._foreach_mul(grads, scale.reshape(()))
            torch._foreach_add_(params, grads, alpha=-lr)

    def zero_dense_grads(self):
        for p in self.dense_parameters():
            p.grad = None


class _ShardedFM(torch.autograd.Function):
    """The 26-field FM over row-sharded tables, with the batch's rows deduplicated before they
    cross the fabric:

      forward   sort + dedup the batch's occurrence list locally; ask the owners for the *distinct*
                rows only (feature + bias, one plan); run rx_fm2_fwd on the received buffer, addressed
                through the inverse map; keep T_b = sum of the sample's rows
      backward  rx_segreduce_fm turns (T, dL/dlogit) into one gradient row per distinct row, with
                no per-occurrence rows; those compact rows (and the bias gradients) go back to the
                owners, which sort what arrived from all ranks and reduce + update.

    Per step a rank sends ~rho * n indices and receives / returns ~rho * n rows instead of n."""

    @staticmethod
    def forward(ctx, anchor, model, flat, B, F, deep_cols=None, extra=None, prep=None):
        """``deep_cols``: also return the tower input [B, deep_cols] (gathered rows [item, field 0 ..
        F-1], then ``extra``, e.g. the dense features); its gradient comes back into the rows'."""
        ops, feat, bias = model.ops, model.feature, model.bias
        n = flat.numel()
        want_deep = deep_cols is not None
        packed = slot_of = occ_slot = None
        if prep is not None:        # prepared on the side stream (ShardedFieldFM.prepare)
            si, packed, slot_of, occ_slot = model._join_prep(prep, flat.device)
        else:
            si = ops.sort_dedup(flat, feat.n_rows, want_inverse=True)
        if model.capacity:
            # nothing here waits for the host.  The rows stay in arrival (slot) order: occurrence ->
            # distinct row -> slot is composed on the index side, 8 bytes per occurrence instead of a
            # copy of every row
            plan = feat.plan_fixed(si.uniq_rows, si.n_uniq, model.capacity, packed)
            if flat.is_cuda and ctx.needs_input_grad[0]:     # a backward pass will follow
                side = model._pos.get("owner_side")
                if side is None:
                    side = model._pos["owner_side"] = torch.cuda.Stream(device=flat.device)
                feat.sort_received_ahead(plan, side)
                with torch.cuda.stream(side):      # the send buffers of the backward pass, off the critical path
                    n_slots = plan["recv_idx"].numel()
                    if feat.dim % 4 == 0:           # the reduce writes every used slot: only the padding slots need zeros
                        ctx.sent = (torch.empty(n_slots, feat.dim, dtype=torch.float32, device=flat.device),
                                    torch.empty(n_slots, dtype=torch.float32, device=flat.device), torch.cuda.Event())
                        plan["counts"].record_stream(side)
                        ops.zero_unused_slots(ctx.sent[0], ctx.sent[1], plan["counts"], model.capacity)
                    else:
                        ctx.sent = (torch.zeros(n_slots, feat.dim, dtype=torch.float32, device=flat.device),
                                    torch.zeros(n_slots, dtype=torch.float32, device=flat.device), torch.cuda.Event())
                    ctx.sent[2].record(side)
                    if model._side_work is not None:     # e.g. a recorded step doing the next batch's index work here
                        work, model._side_work = model._side_work, None
                        work()
            rows_p, bias_p = feat.gather_packed(plan), bias.gather_packed(plan)
            if occ_slot is None:
                slot_of = plan["src_of"].to(torch.int64)
                occ_slot = slot_of[si.inverse.to(torch.int64)]
            base = model._zero_base(F, flat.device)
            logits, deep, tsum = ops.fm2_fwd(rows_p, bias_p.reshape(-1), occ_slot[:B * F].reshape(B, F), base,
                                             occ_slot[B * F:].reshape(B, 1), 0, want_deep=want_deep, deep_cols=deep_cols,
                                             extra=extra, want_tsum=True)
            ctx.model, ctx.plan, ctx.si, ctx.F, ctx.slot_of = model, plan, si, F, slot_of
            ctx.save_for_backward(tsum, rows_p)
            return logits, deep
        else:
            plan = feat.plan(si.uniq_rows, si.n_uniq)    # padded list + device count; one host read inside
            nu = plan["n"]
        rows_u = feat.gather(plan)                        # [nu, D], in distinct-row order
        bias_u = bias.gather(plan)                        # [nu, 1]
        inv = si.inverse.to(torch.int64)
        ex = inv[:B * F].reshape(B, F)
        item = inv[B * F:].reshape(B, 1)
        base = model._zero_base(F, flat.device)
        logits, deep, tsum = ops.fm2_fwd(rows_u, bias_u.reshape(-1), ex, base, item, 0, want_deep=want_deep,
                                         deep_cols=deep_cols, extra=extra, want_tsum=True)
        ctx.model, ctx.plan, ctx.si, ctx.nu, ctx.F, ctx.slot_of = model, plan, si, nu, F, None
        ctx.save_for_backward(tsum, rows_u)
        return logits, deep

    @staticmethod
    def backward(ctx, g, g_deep=None):
        tsum, rows_u = ctx.saved_tensors
        if g is None:
            g = torch.zeros(tsum.shape[0], dtype=torch.float32, device=tsum.device)
        if g_deep is not None and g_deep.stride(-1) != 1:
            g_deep = g_deep.contiguous()
        model, si = ctx.model, ctx.si
        ops, feat, bias = model.ops, model.feature, model.bias
        if ctx.slot_of is not None:
            # fixed capacity: distinct row u lives in slot slot_of[u] of the exchange buffers, and the
            # reduce writes its gradient row straight into the send buffer's slot (unused slots stay 0)
            n_slots = rows_u.shape[0]
            by_slot = ops.SortedIndex(si.n, n_slots, si.sorted_pos, si.sorted_slot, ctx.slot_of, si.seg_off, si.n_uniq, None)
            ahead = getattr(ctx, "sent", None)
            if ahead is not None:       # zero-filled on the side stream during the forward pass
                sent, sent2, filled = ahead
                main = torch.cuda.current_stream(g.device)
                main.wait_event(filled)
                sent.record_stream(main); sent2.record_stream(main)
            else:
                sent = torch.zeros(n_slots, feat.dim, dtype=torch.float32, device=g.device)
                sent2 = torch.zeros(n_slots, dtype=torch.float32, device=g.device)
            ops.segreduce_fm(tsum, g.reshape(-1).contiguous(), ctx.F, by_slot, _lib.RX_SEG_DENSE, sent, table=rows_u,
                             target2=sent2, deep=g_deep)
            feat.reduce_packed(sent, ctx.plan, bias, sent2)
            return (None,) * 8
        nu = ctx.nu
        # the received buffer is addressed by slot, so the reduce sees "row u = slot u"
        slots = ops.SortedIndex(si.n, max(nu, 1), si.sorted_pos, si.sorted_slot, model._iota(si.n, g.device), si.seg_off,
                                si.n_uniq, None)
        compact = torch.empty(max(nu, 1), feat.dim, dtype=torch.float32, device=g.device)
        compact2 = torch.empty(max(nu, 1), dtype=torch.float32, device=g.device)
        ops.segreduce_fm(tsum, g.reshape(-1).contiguous(), ctx.F, slots, _lib.RX_SEG_COMPACT, compact, table=rows_u,
                         target2=compact2, deep=g_deep)
        feat.scatter_grad(compact[:nu], ctx.plan, bias, compact2[:nu])
        return (None,) * 8
