"""HIP-graph capture of the whole train step (forward, loss, backward, fused update).

A step of these models is a few dozen short kernels; launched one by one from Python the host,
not the GPU, sets the pace.  ``GraphedTrainStep`` records the step once per resident batch with
stream capture (torch.cuda.CUDAGraph drives hipStreamBeginCapture; every C-ABI launch goes to the
capturing stream) and replays it with one hipGraphLaunch.  Nothing in the step synchronises with
the host: index errors are reported through the deferred flag (ops.raise_if_index_error), the
loss stays on the device until asked for.

With ``pipeline=True`` (and at least two resident batches) the index preparation of the NEXT batch
(``model.prepare``: flat row ids, radix sort, dedup — work that depends on the indices only) is
recorded on a side stream inside the graph of the CURRENT batch, so it overlaps the current step
instead of sitting on its critical path.  Every replay still performs exactly one such preparation;
replays must then follow the cyclic slot order 0, 1, ..., K-1, 0, ... (``copy_batch`` of slot i+1 has
to happen before slot i is replayed).

``cycle=True`` additionally records graphs that hold SEVERAL consecutive steps: one graph for the
whole round over the resident batches (slots 0..K-1 in order), or, with ``cycle_len`` < K, one graph
per segment of ``cycle_len`` consecutive slots (the last segment may be shorter).  Replaying one
(``run_cycle``) removes the host's launch gap between steps, and the index preparation runs ahead on
its side stream: step i only waits for the preparation of slot i; the preparation of a slot that the
same graph has already consumed (a whole-round graph preparing slot 0 again) waits for that step.
A segment must be replayed when its first slot is next in the cyclic order.

Limits: the batch tensors are captured by address, so a graph is tied to its batch object
(``add_batch`` records one graph per resident batch; ``copy_batch`` refreshes a batch in place).
Dense Adam keeps its step count in device memory and advances it inside the recording.
"""
from __future__ import annotations

import os
from typing import Dict, List

import torch

from . import ops
from .autograd import backward_from_loss, clear_capture_sort_cache, clear_sort_cache


def _prep_stream(dev) -> "torch.cuda.Stream":
    """The one stream per device on which recordings fork the next batch's index preparation (a new
    stream per recording would leave the process with as many HIP streams as it has graphs, all
    multiplexed onto the device's few hardware queues)."""
    from .autograd import _side_stream_for
    return _side_stream_for("prep", dev)


_MAX_WIDTH = int(os.environ.get("RX_GRAPH_MAX_WIDTH", 2))
# Multi-step recordings: the index preparation of slot i + 1 is tied to the start of step i.  Left free to run
# ahead (RX_GRAPH_FREE_RUNNING=1, the earlier form) the replayed graph does not overlap the two branches as
# recorded: the kernel trace shows the side branch's ten preparations back to back with the main branch stalled
# behind them for most of that time, then the ten steps alone.
_LOCKSTEP = os.environ.get("RX_GRAPH_FREE_RUNNING") != "1"
# (Also tried: a step waiting for its slot's preparation only where the backward's reduce first reads it, so that its
# forward pass overlaps the preparation's tail.  0.133 against 0.127 ms per step on the 26-field FM: the forward and
# the grouping's scatter are both bound by memory, and side by side each slows the other by more than the wait cost.)
_hip = None


def recorded_shape(g: "torch.cuda.CUDAGraph") -> dict:
    """Nodes, edges, roots and the widest level (longest-path layering: the nodes of one level cannot
    reach one another, so a launch may run that many at once) of a recording made with
    ``keep_graph=True``, from the runtime's own queries (hipGraphGetNodes / hipGraphGetEdges)."""
    global _hip
    import collections
    import ctypes as C
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
    raw = C.c_void_p(g.raw_cuda_graph())
    n = C.c_size_t(0)
    if _hip.hipGraphGetNodes(raw, None, C.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    nodes = (C.c_void_p * max(n.value, 1))()
    _hip.hipGraphGetNodes(raw, nodes, C.byref(n))
    ne = C.c_size_t(0)
    if _hip.hipGraphGetEdges(raw, None, None, C.byref(ne)) != 0:
        raise RuntimeError("hipGraphGetEdges failed")
    src, dst = (C.c_void_p * max(ne.value, 1))(), (C.c_void_p * max(ne.value, 1))()
    _hip.hipGraphGetEdges(raw, src, dst, C.byref(ne))
    succ, indeg = collections.defaultdict(list), collections.Counter()
    for i in range(ne.value):
        succ[src[i]].append(dst[i])
        indeg[dst[i]] += 1
    names = [nodes[i] for i in range(n.value)]
    level = {x: 0 for x in names if indeg[x] == 0}
    roots, todo = len(level), list(level)
    while todo:
        x = todo.pop()
        for y in succ[x]:
            level[y] = max(level.get(y, 0), level[x] + 1)
            indeg[y] -= 1
            if indeg[y] == 0:
                todo.append(y)
    per_level = collections.Counter(level.values())
    return {"nodes": n.value, "edges": ne.value, "roots": roots, "widest_level": max(per_level.values(), default=0)}


def _new_graph() -> "torch.cuda.CUDAGraph":
    return torch.cuda.CUDAGraph(keep_graph=True) if _MAX_WIDTH > 0 else torch.cuda.CUDAGraph()


def _check_width(g: "torch.cuda.CUDAGraph", what: str) -> None:
    """Refuse a recording in which more than ``RX_GRAPH_MAX_WIDTH`` (default 2) kernels can run at once.
    Replays of three-wide recordings of these steps (a valid DAG of kernel nodes only, one edge fewer
    than its two-wide form) ended in a host segmentation fault inside hipGraphLaunch now and then on
    this runtime (4 hardware queues per process); two-wide ones have not.  0 switches the check off."""
    if _MAX_WIDTH <= 0:
        return
    shape = recorded_shape(g)
    if shape["widest_level"] > _MAX_WIDTH:
        raise RuntimeError(f"{what}: the recorded step has {shape['widest_level']} concurrent branches "
                           f"({shape['nodes']} nodes, {shape['roots']} roots); at most {_MAX_WIDTH} are replayed "
                           "(RX_GRAPH_MAX_WIDTH)")
    g.instantiate()


def refuse_live_autograd_graphs(model) -> None:
    """A step must not be recorded while an autograd graph of an EARLIER step over these parameters is
    still alive (typically a loss tensor of an eager step that was kept in a variable).  Such a graph
    keeps the parameters' gradient-accumulation nodes alive, and those nodes stay bound to the stream
    they were made on -- for an eager step the legacy default stream.  The recorded backward pass
    would then run them there, which drags the default stream into the capture; on this runtime
    that ends in a host segmentation fault inside hipStreamEndCapture (reproduced with dense table
    gradients: profiles/stale_autograd_probe.py).  Detected here instead: a mark left on a
    parameter's accumulation node survives only if something else holds that node."""
    if os.environ.get("RX_ALLOW_LIVE_AUTOGRAD") == "1":      # profiles/stale_autograd_probe.py only
        return
    token = object()
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        torch.autograd.graph.get_gradient_edge(p).node.metadata["rx_live_probe"] = token
        if torch.autograd.graph.get_gradient_edge(p).node.metadata.get("rx_live_probe") is token:
            torch.autograd.graph.get_gradient_edge(p).node.metadata.pop("rx_live_probe", None)
            raise RuntimeError(f"cannot record the train step: an autograd graph of an earlier step still references "
                               f"parameter {name!r} (a loss tensor kept alive?); delete or detach it first")


class GraphedTrainStep:
    def __init__(self, model, optimizer, warmup: int = 3, pipeline: bool = False, cycle: bool = False,
                 cycle_len: int = 0, ahead: int = 1):
        """``ahead`` (with ``pipeline``): the step on slot i is recorded beside the index preparation of slot
        i + ahead.  With 1 a step starts when the preparation that ran beside the previous step has finished, so
        the longer of the two chains sets the pace and the shorter one idles at every join; with 2 a step's
        preparation finished a whole step earlier and the chains only meet when one falls a full step behind.
        A refreshed slot (``copy_batch``) must be in place before the step ``ahead`` slots earlier is launched."""
        self.model, self.opt, self.warmup = model, optimizer, warmup
        self.ahead = max(1, int(ahead))
        self.graphs: List[torch.cuda.CUDAGraph] = []
        self.batches: List[Dict[str, torch.Tensor]] = []
        self.losses: List[torch.Tensor] = []
        self._pool = None
        self.pipeline = pipeline
        self.cycle = cycle
        self.cycle_len = cycle_len
        self.cycle_graph = None                       # the graph of the segment that starts at slot 0
        self.cycle_losses: List[torch.Tensor] = []
        self.segments: Dict[int, tuple] = {}          # first slot -> (graph, number of steps, losses)

    def _eager(self, batch):
        if torch.cuda.is_current_stream_capturing():
            # a recording of several steps refills its static batches through raw pointers between the steps
            # (no version bump), so a sort memoised by an earlier step of the recording must not serve this one
            clear_capture_sort_cache()
        self.opt.zero_grad()
        from . import autograd as rxa
        rxa.DEFER_LOSS_FINAL[0] = True      # forward, backward and step follow one another: the reduce may write the loss
        try:
            loss = self.model.loss(batch)
            backward_from_loss(loss)
        finally:
            rxa.DEFER_LOSS_FINAL[0] = False
        self.opt.step()
        return loss

    def add_batches(self, batches) -> None:
        """Register all resident batches, then record their graphs (needed for ``pipeline=True``,
        where the graph of slot i also prepares slot i+1)."""
        batches = list(batches)
        if not (self.pipeline and len(batches) >= 2):
            for b in batches:
                self.add_batch(b)
            return
        if ops._INDEX_CHECK != "deferred":
            raise RuntimeError("graph capture needs ops.set_index_check('deferred'): a captured step cannot sync")
        dev = next(self.model.parameters()).device
        refuse_live_autograd_graphs(self.model)
        prepared = [self.model.prepare(b) for b in batches]          # static buffers, filled eagerly once
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self._eager(prepared[0])
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        K = len(prepared)
        A = self.ahead = min(self.ahead, K - 1)
        shared = self._shared_branch(dev)
        for i, b in enumerate(prepared):
            g = _new_graph()
            self.opt.zero_grad()
            clear_sort_cache()
            with torch.cuda.graph(g, pool=self._pool):
                main = torch.cuda.current_stream(dev)
                if shared is None:
                    fork = _prep_stream(dev)
                    fork.wait_stream(main)
                    with torch.cuda.stream(fork):
                        self.model.prepare(prepared[(i + A) % K], out=prepared[(i + A) % K])
                    loss = self._eager(b)
                else:       # behind the model's own branch work, on its stream (see _shared_branch)
                    fork = shared
                    fork.wait_stream(main)
                    loss = self._eager(b)
                    with torch.cuda.stream(fork):
                        self.model.prepare(prepared[(i + A) % K], out=prepared[(i + A) % K])
                main.wait_stream(fork)
            _check_width(g, "GraphedTrainStep.add_batches")
            if self._pool is None:
                self._pool = g.pool()
            self.graphs.append(g)
            self.batches.append(b)
            self.losses.append(loss.detach())      # detached: no autograd graph outlives its recording
            del loss
        if self.cycle:
            self._record_cycle(prepared, dev)

    def _shared_branch(self, dev):
        """Models that already run part of their step on a second stream (``_branch_streams``: NeuMF's GMF
        half, autograd.Branch) get the next batch's index preparation on that same stream, behind the
        branch's own work of the step, instead of on a forked stream of its own: a recorded step then has
        two concurrent branches, not three.  With three, replays crashed inside the HIP runtime now and
        then (a host segmentation fault in hipGraphLaunch)."""
        if not getattr(self.model, "_branch_streams", False) or os.environ.get("RX_GRAPH_OWN_FORK") == "1":
            return None     # (RX_GRAPH_OWN_FORK=1: the three-branch recording, for profiles/graph_shape_probe.py only)
        from .autograd import _side_stream_for
        return _side_stream_for("branch", dev)

    def _record_cycle(self, prepared, dev) -> None:
        K = len(prepared)
        shared = self._shared_branch(dev)
        seg = self.cycle_len if 0 < self.cycle_len < K else K
        for first in range(0, K, seg):
            slots = list(range(first, min(first + seg, K)))
            g = _new_graph()
            losses: List[torch.Tensor] = []
            self.opt.zero_grad()
            clear_sort_cache()
            with torch.cuda.graph(g, pool=self._pool):
                main = torch.cuda.current_stream(dev)
                fork = shared if shared is not None else _prep_stream(dev)
                fork.wait_stream(main)
                ready = {}            # slot -> event behind its preparation inside this graph
                done = {}             # slot -> event recorded after this graph's step on it

                def prepare_next(nxt):
                    if _LOCKSTEP and shared is None:
                        # the preparation starts where the step beside it starts (behind the previous step's last
                        # launch) instead of running ahead freely: see _LOCKSTEP
                        fork.wait_stream(main)
                    with torch.cuda.stream(fork):
                        if nxt in done:
                            fork.wait_event(done[nxt])   # its sort result was read by a step of this graph
                        self.model.prepare(prepared[nxt], out=prepared[nxt])
                        ev = torch.cuda.Event()
                        ev.record(fork)
                    return ev

                for i in slots:
                    if i in ready:
                        main.wait_event(ready.pop(i))
                    nxt = (i + self.ahead) % K
                    if shared is None:
                        ready[nxt] = prepare_next(nxt)
                    losses.append(self._eager(prepared[i]).detach())
                    done[i] = torch.cuda.Event()
                    done[i].record(main)
                    if shared is not None:      # behind this step's branch work on the branch's stream
                        ready[nxt] = prepare_next(nxt)
                main.wait_stream(fork)
            _check_width(g, "GraphedTrainStep (multi-step segment)")
            self.segments[first] = (g, len(slots), losses)
        self.cycle_graph = self.segments[0][0]
        self.cycle_losses = self.segments[0][2]

    def segment_at(self, slot: int) -> int:
        """Number of steps of the multi-step graph that starts at ``slot`` (0: none starts there)."""
        entry = self.segments.get(slot)
        return entry[1] if entry else 0

    def run_cycle(self, first: int = 0) -> List[torch.Tensor]:
        """The steps of the segment that starts at slot ``first`` (the whole round when there is one
        segment), in slot order, as a single graph launch."""
        entry = self.segments.get(first)
        if entry is None:
            raise RuntimeError("no multi-step graph starts at this slot: construct with pipeline=True, cycle=True and >= 2 batches")
        entry[0].replay()
        return entry[2]

    def add_batch(self, batch: Dict[str, torch.Tensor], epilogue=None) -> int:
        """Record one graph bound to ``batch``'s tensors.  Before the first recording the step is
        run eagerly ``warmup`` times on a side stream (these are real train steps on ``batch``);
        recording itself executes nothing.  ``epilogue(loss)``: further launches recorded behind the
        step (utils.LoopGraph records the hand-over to the next batch there)."""
        if ops._INDEX_CHECK != "deferred":
            raise RuntimeError("graph capture needs ops.set_index_check('deferred'): a captured step cannot sync")
        dev = next(self.model.parameters()).device
        refuse_live_autograd_graphs(self.model)
        if not self.graphs:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(self.warmup):
                    self._eager(batch)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
        g = _new_graph()
        self.opt.zero_grad()
        clear_sort_cache()   # the recording must hold its own sort kernels (see autograd.sorted_index)
        with torch.cuda.graph(g, pool=self._pool):
            loss = self._eager(batch).detach()
            if epilogue is not None:
                epilogue(loss)
        _check_width(g, "GraphedTrainStep.add_batch")
        clear_sort_cache()   # memo entries made while recording point into the graph's pool
        if self._pool is None:
            self._pool = g.pool()
        self.graphs.append(g)
        self.batches.append(batch)
        self.losses.append(loss)
        return len(self.graphs) - 1

    def add_pair(self, batches, epilogues, unroll: int = 8):
        """Two static batches that take turns (utils.LoopGraph over a resident epoch): the graph of
        slot s runs the step on batch s, prepares the indices of batch 1 - s on a forked stream beside
        it, and ends with ``epilogues[s](loss)`` (the hand-over that refills batch s).  Replays must
        alternate s = 0, 1, 0, ...; before the first replay of slot 0 its preparation has to be current
        (``refresh_prepared(0)``).  A third graph holds ``unroll`` (even) such steps starting at slot 0
        (``run_pair_steps``): one host launch per ``unroll`` steps, for steps short enough that the
        launch itself sets the pace.  The recording runs the index preparation of both batches once and
        nothing else.  Returns the two slot numbers."""
        if ops._INDEX_CHECK != "deferred":
            raise RuntimeError("graph capture needs ops.set_index_check('deferred'): a captured step cannot sync")
        dev = next(self.model.parameters()).device
        refuse_live_autograd_graphs(self.model)
        prepared = [self.model.prepare(b) for b in batches]      # static preparation buffers (filled eagerly once)

        def record(order):
            g = _new_graph()
            self.opt.zero_grad()
            clear_sort_cache()
            with torch.cuda.graph(g, pool=self._pool):
                main = torch.cuda.current_stream(dev)
                shared = self._shared_branch(dev)
                fork = shared if shared is not None else _prep_stream(dev)     # one side stream for all steps of the recording
                for s in order:
                    fork.wait_stream(main)       # behind the hand-over that refilled batch 1 - s
                    if shared is None:
                        with torch.cuda.stream(fork):
                            self.model.prepare(prepared[1 - s], out=prepared[1 - s])
                    loss = self._eager(prepared[s]).detach()
                    if shared is not None:       # behind the model's own branch work of this step (see _shared_branch)
                        with torch.cuda.stream(fork):
                            self.model.prepare(prepared[1 - s], out=prepared[1 - s])
                    epilogues[s](loss)
                    main.wait_stream(fork)
            _check_width(g, "GraphedTrainStep.add_pair")
            clear_sort_cache()
            if self._pool is None:
                self._pool = g.pool()
            return g, loss

        slots = []
        for s in (0, 1):
            g, loss = record([s])
            self.graphs.append(g)
            self.batches.append(batches[s])
            self.losses.append(loss)
            slots.append(len(self.graphs) - 1)
        unroll -= unroll % 2
        self._pair_multi = (record([k % 2 for k in range(unroll)])[0], unroll) if unroll >= 4 else None
        self._pair_prepared, self._pair_slots = prepared, slots
        return slots

    def run_pair_steps(self, count: int) -> None:
        """``count`` alternating steps of the pair starting at its slot 0: multi-step graph launches
        while enough steps remain, single-step graphs for the rest."""
        done = 0
        if self._pair_multi is not None:
            g, u = self._pair_multi
            while count - done >= u:
                g.replay()
                done += u
        while done < count:
            self.graphs[self._pair_slots[done % 2]].replay()
            done += 1

    def refresh_prepared(self, s: int) -> None:
        """Index preparation of the pair's static batch ``s`` from its current contents (eager)."""
        self.model.prepare(self._pair_prepared[s], out=self._pair_prepared[s])

    def copy_batch(self, slot: int, new_batch: Dict[str, torch.Tensor]) -> None:
        for k, v in self.batches[slot].items():
            if k in new_batch:
                v.copy_(new_batch[k])

    def __call__(self, slot: int) -> torch.Tensor:
        self.graphs[slot].replay()
        return self.losses[slot]


class ShardedGraphStep:
    """One HIP graph per resident batch for a row-sharded model (``sharded.ShardedFieldFM`` and its
    DeepFM variant) running its fixed-capacity exchange, where no size crosses to the host: the
    whole step -- index work, all-to-all exchanges (RCCL calls are recorded like kernels), forward,
    backward, owner-side reduce, clip, update -- replays as one launch.

    ``step_fn(batch) -> loss`` runs one complete train step eagerly; it is called ``warmup`` times on
    the first batch before recording (real steps).  Every rank must construct, replay and release
    these graphs in the same order.  Call ``release()`` before ``destroy_process_group()``: tearing
    the group down while recorded collectives are alive blocks."""

    def __init__(self, step_fn, batches, device, warmup: int = 3, model=None):
        """``model``: the sharded model, when its batch-only index work (``prepare_static``: occurrence
        list, sort + dedup, owner buckets) should leave the critical path: the graph of batch i then
        does that work for batch i+1 on the forward pass's side stream, into buffers the graph of batch i+1 reads,
        instead of doing its own up front.  Needs at least two batches; replays must then follow the
        batch order (``slot``, ``slot + 1``, ... cyclically), as every driver here does."""
        if ops._INDEX_CHECK != "deferred":
            raise RuntimeError("graph capture needs ops.set_index_check('deferred'): a captured step cannot sync")
        self.batches = list(batches)
        self.graphs: List[torch.cuda.CUDAGraph] = []
        self.losses: List[torch.Tensor] = []
        self.segments: Dict[int, tuple] = {}      # no multi-step graphs (same driver interface as GraphedTrainStep)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step_fn(self.batches[0])
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        pool = None
        self.ahead = model is not None and len(self.batches) >= 2 and bool(getattr(model, "capacity", None)) \
            and hasattr(model, "prepare_static")
        if self.ahead:
            self.prepared = [model.prepare_static(b) for b in self.batches]      # static buffers, filled eagerly once
            torch.cuda.synchronize(device)
        K = len(self.batches)
        for i, b in enumerate(self.batches):
            g = _new_graph()
            clear_sort_cache()
            # thread-local error mode: the process group's watchdog thread polls the events of earlier
            # collectives while this thread records, which the global mode turns into a capture error
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                if self.ahead:
                    # the next batch's index work rides on the stream the forward pass forks for its
                    # owner-side sort (after that sort and the send-buffer fill): two concurrent
                    # branches in the graph, not three
                    nxt = (i + 1) % K
                    model._side_work = lambda: model.prepare_static(self.batches[nxt], out=self.prepared[nxt])
                    try:
                        loss = step_fn(self.prepared[i])
                        taken = model._side_work is None
                    finally:
                        model._side_work = None               # never left behind for a later, unrecorded forward pass
                    rec = torch.cuda.current_stream(device)
                    if taken:
                        rec.wait_stream(model._pos["owner_side"])                 # join before the graph ends
                    else:                                     # no forward pass took it: do it here
                        model.prepare_static(self.batches[nxt], out=self.prepared[nxt])
                else:
                    loss = step_fn(b)
            _check_width(g, "ShardedGraphStep")
            clear_sort_cache()
            pool = pool or g.pool()
            self.graphs.append(g)
            self.losses.append(loss.detach())

    def __call__(self, slot: int) -> torch.Tensor:
        self.graphs[slot].replay()
        return self.losses[slot]

    def segment_at(self, slot: int) -> int:
        return 0          # one graph per step (same driver interface as GraphedTrainStep)

    def release(self) -> None:
        self.graphs.clear()
        self.losses.clear()
        self.prepared = []
