#!/usr/bin/env python3
"""Train-step throughput of the embedding + interaction hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A *step* is one pass of fwd + bwd + update over one batch of seeded synthetic input that is
already resident in HBM.  A *sample* is one batch row.  The default workload is the
Criteo-shaped 26-field factorization machine (D = 16, batch 65536 per GPU): the configuration
on which "embedding HBM GB/s vs peak" is meaningful.  The step follows the reference's train
step (recsys/utils.py:96-103: zero_grad, loss, backward, global grad-norm clip at 1.0,
optimizer step) with the optimizer being the fused sparse SGD row update.

Rank 0 prints ONE JSON line.  ``roofline`` prices the dominant kernel of the step by its
algorithmic bytes (formulas in algo_bytes below) over its average duration, measured
with HIP events on the launch stream, re-issuing the op back to back with the arguments of its last call
in the step.  ``cpu_baseline`` is the CPU oracle
(a PyTorch restatement of the reference modules and train step) timed on this host's cores on a
bounded number of steps of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


# ----------------------------------------------------------------------------- workloads
def build_workload(name: str, device, rank: int, world: int, clip, use_graph: bool = True):
    """Returns dict(model, opt, batches, samples_per_step, desc, algo_bytes fn)."""
    from recsys_amd import synthetic
    from recsys_amd.models import fields
    from recsys_amd.models.matrix_factorization import VanillaMF
    from recsys_amd.models.neuralcf import NeuMF
    from recsys_amd.optim import FusedOptimizer

    torch.manual_seed(1234)
    # resident batches: enough that the rows a round touches exceed the 256 MiB Infinity Cache several
    # times over for the D = 16 tables (64 batches name about 1 GB of distinct rows), so that the
    # step's table traffic comes from HBM; RX_BENCH_BATCHES overrides (10 = a working set that fits the Infinity Cache)
    n_batches = int(os.environ.get("RX_BENCH_BATCHES", 64 if name in ("fm_criteo",) else 10))
    if name in ("deepfm_criteo_sharded", "deepfm_wide_d128_sharded"):
        # the Criteo-shaped DeepFM with its tables row-sharded and its BatchNorm tower data parallel
        # (tower gradients summed over the ranks by one all-reduce); 33 M rows x 16 per GPU.
        # deepfm_wide_d128_sharded: the 1 B-row x 128 configuration as named -- 125 M rows x 128 per
        # GPU and the 3344-3344-1672-836-1 tower the reference's width rule gives (bf16 operands)
        from recsys_amd.sharded import ShardedFieldDeepFM
        wide = name == "deepfm_wide_d128_sharded"
        dims = synthetic.criteo_field_dims(total_rows=(int(os.environ.get("RX_BENCH_SHARDED_ROWS", 125_000_000)) if wide else 33_000_000) * world)
        B, D, lr = 65536, 128 if wide else 16, 0.05
        if wide:
            n_batches = min(n_batches, 4)
        model = ShardedFieldDeepFM(dims, D, device, bf16=wide)
        model.set_rule("sgd", lr, clip_norm=clip)
        model.train()
        if os.environ.get("RX_BENCH_FORCE_DIST") == "1":
            model.feature.force_exchange = model.bias.force_exchange = True
        batches = [synthetic.criteo_batch(dims, B, seed=1000 * rank + i, device=device) for i in range(n_batches)]
        cap = model.calibrate(batches, slack=1.1)
        desc = {"workload": name, "fields": len(dims), "rows": int(sum(dims)), "dim": D, "batch_per_gpu": B,
                "items_per_row": 1, "optimizer": "fused sparse SGD at the owning rank; tower: all-reduce + dense SGD",
                "clip_norm": clip, "launch": f"eager, asynchronous ({cap} exchange slots per owner)"}

        class _DeepOpt:
            def zero_grad(self):
                model.zero_dense_grads()

            def step(self):
                model.allreduce_dense_grads()
                model.dense_sgd_step(lr, model.finish_step(model.dense_grads()))

        w = dict(model=model, opt=_DeepOpt(), batches=batches, B=B, desc=desc, graphed=None, loss_scale=1.0 / world, dims=dims)
        return graph_sharded(w, device, world) if use_graph else w
    if name == "fm_criteo_sharded":
        # tables row-sharded over the ranks (row r on rank r % G), batch data parallel, index / row /
        # gradient exchange by all-to-all; per-GPU batch fixed (weak scaling)
        from recsys_amd.sharded import ShardedFieldFM
        # weak scaling towards the 1 B-row x 128 configuration: 125 M rows (64 GB of fp32 rows) per GPU,
        # same cardinality profile; at 8 GPUs the tables total 1.0 B rows
        rows_per_gpu = int(os.environ.get("RX_BENCH_SHARDED_ROWS", 125_000_000))
        dims = synthetic.criteo_field_dims(total_rows=rows_per_gpu * world)
        B, D = 65536, int(os.environ.get("RX_BENCH_SHARDED_DIM", 128))
        model = ShardedFieldFM(dims, D, device)
        model.set_rule("sgd", 0.05, clip_norm=clip)
        if os.environ.get("RX_BENCH_FORCE_DIST") == "1":
            model.feature.force_exchange = model.bias.force_exchange = True
        batches = [synthetic.criteo_batch(dims, B, seed=1000 * rank + i, device=device) for i in range(n_batches)]
        cap = model.calibrate(batches, slack=1.1)   # fixed-capacity exchange: equal all-to-all splits, no host read in the step
        desc = {"workload": name, "fields": len(dims), "rows": int(sum(dims)), "dim": D, "batch_per_gpu": B,
                "items_per_row": 1, "optimizer": "fused sparse SGD at the owning rank", "clip_norm": clip,
                "launch": f"eager, asynchronous (distinct rows only, {cap} exchange slots per owner)"}

        class _Opt:
            def zero_grad(self):
                pass

            def step(self):
                model.finish_step()     # global-norm clip over all ranks, then the pending row updates

        # the global loss is the mean over all ranks' rows: each rank back-propagates loss / G
        w = dict(model=model, opt=_Opt(), batches=batches, B=B, desc=desc, graphed=None, loss_scale=1.0 / world, dims=dims)
        return graph_sharded(w, device, world) if use_graph else w
    if name in ("fm_criteo", "fm_wide_d128", "deepfm_criteo", "deepfm_criteo_bf16", "dcnv2_criteo"):
        dims = synthetic.criteo_field_dims()
        B, D = 65536, int(os.environ.get("RX_BENCH_DIM", 16))      # RX_BENCH_DIM: row-width sweeps
        if name == "fm_wide_d128":
            # one GPU's share of the 1 B-row x 128 configuration, unsharded: 125 M rows of 512 bytes (64 GB)
            dims, D = synthetic.criteo_field_dims(total_rows=125_000_000), 128
        cls = {"fm_criteo": fields.FieldFM, "fm_wide_d128": fields.FieldFM, "deepfm_criteo": fields.FieldDeepFM,
               "deepfm_criteo_bf16": fields.FieldDeepFM, "dcnv2_criteo": fields.DCNv2}[name]
        # fm_criteo: rows of 16 floats sit in 128-byte lines of their own with the bias entry behind them
        # (fields._FieldTable: same parameters and results, other strides); RX_BENCH_NO_LINE_ROWS=1 for A/B
        line_rows = name == "fm_criteo" and D == 16 and os.environ.get("RX_BENCH_NO_LINE_ROWS") != "1"
        with torch.device(device):
            if name == "deepfm_criteo_bf16":
                model = cls(dims, D, bf16=True)
            else:
                model = cls(dims, D, line_rows=True) if line_rows else cls(dims, D)
        batches = [synthetic.criteo_batch(dims, B, seed=1000 * rank + i, device=device) for i in range(n_batches)]
        desc = {"workload": name, "fields": len(dims), "rows": int(sum(dims)), "dim": D, "batch_per_gpu": B,
                "items_per_row": 1, "optimizer": "fused sparse SGD", "clip_norm": clip}
        if line_rows:
            desc["table_layout"] = "one 128-byte line per row: 16 floats, the bias entry, zeros"
        lr = 0.05
    elif name in ("mf_ml1m", "neumf_ml1m"):
        B = 1024 if name == "mf_ml1m" else 8192
        D = 64
        with torch.device(device):
            if name == "mf_ml1m":
                model = VanillaMF(synthetic.ML1M["n_users"], synthetic.ML1M["n_items"], D)
            else:
                model = NeuMF(synthetic.ML1M["n_users"], synthetic.ML1M["n_items"], D)
        batches = [synthetic.ml1m_batch(B, seed=1000 * rank + i, device=device) for i in range(n_batches)]
        desc = {"workload": name, "rows": synthetic.ML1M["n_users"] + synthetic.ML1M["n_items"], "dim": D,
                "batch_per_gpu": B, "items_per_row": 5, "optimizer": "fused sparse SGD", "clip_norm": clip}
        lr = 0.05
    else:
        raise SystemExit(f"unknown workload {name}")
    model.train()
    opt = FusedOptimizer(model, kind="sgd", lr=lr, clip_norm=clip)
    w = dict(model=model, opt=opt, batches=batches, B=B, desc=desc, graphed=None, dims=dims if "fields" in desc else None)
    if use_graph:
        from recsys_amd.graph import GraphedTrainStep
        gs = GraphedTrainStep(model, opt, pipeline=True, cycle=os.environ.get("RX_BENCH_NO_CYCLE") != "1", cycle_len=10,
                              # preparation depth, measured per workload (A/B on one box): two slots ahead where the side
                              # chain is as long as the step (fm_criteo 0.1265 -> 0.1237 ms, neumf_ml1m 0.095 -> 0.080), one
                              # where it is short (mf_ml1m 0.0525 against 0.0556 with two, deepfm_criteo_bf16 1.349 / 1.357)
                              ahead=int(os.environ.get("RX_BENCH_PREP_AHEAD", 2 if name in ("fm_criteo", "neumf_ml1m") else 1)))
        gs.add_batches(batches)
        w["graphed"] = gs
        desc["resident_batches"] = n_batches
        desc["launch"] = (f"HIP graph replays over {n_batches} resident batches: one graph per 10 consecutive steps where "
                          f"the step count allows, one graph per step otherwise; the indices of batch i+{gs.ahead} are grouped on a "
                          "side stream while batch i trains (one index preparation per step, off the critical path)")
    else:
        desc["launch"] = "eager"
    return w


def graph_sharded(w, device, world):
    """Sharded workloads: one HIP graph per resident batch (index work, exchanges, backward, clip and
    update in one launch).  On by default at one rank, where it is tested against eager steps bit for
    bit with the RCCL calls inside the recording; at more ranks eager launch unless
    RX_BENCH_SHARDED_GRAPH=1 (recorded collectives could not be rehearsed on several GPUs here)."""
    want = os.environ.get("RX_BENCH_SHARDED_GRAPH")
    if want == "0" or (world > 1 and want != "1"):
        return w
    from recsys_amd.graph import ShardedGraphStep
    model, opt, scale = w["model"], w["opt"], w["loss_scale"]

    def step(batch):
        opt.zero_grad()
        loss = model.loss(batch)
        (loss * scale).backward()
        opt.step()
        return loss.detach()

    # RX_BENCH_SHARDED_NO_AHEAD=1: every graph does its own batch's index work up front (A/B)
    w["graphed"] = ShardedGraphStep(step, w["batches"], device, warmup=3,
                                    model=None if os.environ.get("RX_BENCH_SHARDED_NO_AHEAD") == "1" else model)
    w["desc"]["launch"] = w["desc"]["launch"].replace("eager, asynchronous", "one HIP graph per step (exchanges recorded inside)")
    return w


def run_steps(w, first: int, count: int):
    """Steps first .. first+count-1 (batch = step % resident batches); whole rounds go through the
    round graph when there is one.  Returns the last loss."""
    gs, K = w.get("graphed"), len(w["batches"])
    i, end, loss = first, first + count, None
    while i < end:
        seg = gs.segment_at(i % K) if gs is not None else 0
        if seg and end - i >= seg:
            loss = gs.run_cycle(i % K)[-1]
            i += seg
        else:
            loss = one_step(w, i)
            i += 1
    return loss


def one_step(w, i):
    if w.get("graphed") is not None:
        return w["graphed"](i % len(w["batches"]))
    batch = w["batches"][i % len(w["batches"])]
    if hasattr(w["model"], "prepare") and w.get("loss_scale"):
        # sharded models: the local index work of the next batch goes to a side stream before this
        # step is launched, and overlaps it
        ready = w.setdefault("prepared", {})
        batch = ready.pop(i, None) or w["model"].prepare(batch)
        ready[i + 1] = w["model"].prepare(w["batches"][(i + 1) % len(w["batches"])])
    w["opt"].zero_grad()
    loss = w["model"].loss(batch)
    if w.get("loss_scale"):
        (loss * w["loss_scale"]).backward()
    else:
        from recsys_amd.autograd import backward_from_loss
        backward_from_loss(loss)
    w["opt"].step()
    return loss


def self_check(w, pos):
    """Outside the timed regions: the next multi-step graph of the timed launch structure (or one
    single-step graph) against EAGER steps on the same batches from the same weights -- own index
    preparation, one launch per kernel, no replay.  Weights, buffers and the last loss must come out
    bit for bit equal.  Returns (verdict, position after the check)."""
    from recsys_amd.graph import GraphedTrainStep
    gs, K, model = w.get("graphed"), len(w["batches"]), w["model"]
    if not isinstance(gs, GraphedTrainStep):
        return "skipped: the timed steps were not graph replays of a single-GPU step", pos
    if any(isinstance(m, torch.nn.Dropout) and m.p > 0 for m in model.modules()):
        return "skipped: dropout masks of a replay and of an eager step come from different generator states", pos
    while gs.segments and not gs.segment_at(pos % K):
        run_steps(w, pos, 1)
        pos += 1
    n = gs.segment_at(pos % K) or 1
    state = list(model.parameters()) + list(model.buffers())
    torch.cuda.synchronize()
    saved = [t.detach().clone() for t in state]
    loss_g = run_steps(w, pos, n).detach().clone()
    after = [t.detach().clone() for t in state]
    with torch.no_grad():
        for t, s0 in zip(state, saved):
            t.copy_(s0)
    for j in range(n):
        loss_e = gs._eager(w["batches"][(pos + j) % K])      # the raw batch: no prepared index lists
    torch.cuda.synchronize()
    bad = [i for i, (t, a) in enumerate(zip(state, after)) if not torch.equal(t.detach(), a)]
    if bad or not torch.equal(loss_e.detach(), loss_g):
        return f"FAILED: {n} replayed steps and {n} eager steps disagree (tensors {bad}, loss {float(loss_g)} vs {float(loss_e)})", pos + n
    return f"ok ({n} steps of one graph launch == {n} eager steps, bitwise: {len(state)} tensors + loss)", pos + n


# ----------------------------------------------------------------------------- per-op timing
class OpTimer:
    """Wraps chosen functions of recsys_amd.ops with HIP events recorded on the launch stream."""

    def __init__(self, names):
        from recsys_amd import ops
        self.ops, self.names, self.events, self.saved = ops, list(names), {n: [] for n in names}, {}
        self.last_call = {}

    def __enter__(self):
        for n in self.names:
            fn = getattr(self.ops, n)
            self.saved[n] = fn

            def wrapped(*a, __fn=fn, __n=n, **k):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                out = __fn(*a, **k)
                e.record()
                self.events[__n].append((s, e))
                self.last_call[__n] = (a, k)
                return out

            setattr(self.ops, n, wrapped)
        return self

    def __exit__(self, *exc):
        for n, fn in self.saved.items():
            setattr(self.ops, n, fn)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for n, evs in self.events.items():
            if evs:
                ms = [s.elapsed_time(e) for s, e in evs]
                out[n] = (sum(ms) / len(ms), len(ms))
        return out


def back_to_back_ms(op_name, call, reps=30):
    """Average duration of one launch of `op_name`, re-issued `reps` times back to back with the
    arguments of its last call in the step, bracketed by two HIP events on the launch stream.
    Queued launches hide the host's launch latency, so this is the kernels' own time."""
    from recsys_amd import ops
    fn = getattr(ops, op_name)
    args, kwargs = call
    for _ in range(3):
        fn(*args, **kwargs)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn(*args, **kwargs)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps, reps


ALL_OPS = ["fm2_fwd", "fm2_fwd_bce", "fm2_bwd", "fm2_flat_indices", "sort_dedup", "fm_sort_dedup", "fg_group", "segreduce", "segreduce_fm", "fg_reduce",
           "row_update", "fg_row_update", "bce_with_logits",
           "mf_score_fwd", "mf_score_bwd", "gather_concat", "gather_concat_bwd", "gmf_mul_fwd", "gmf_mul_bwd",
           "linear_fwd", "linear_bwd_x", "linear_bwd_w", "relu_bwd", "sumsq_accumulate", "sum_accumulate",
           "clip_scale", "reduce_sum", "cross_combine_fwd", "cross_combine_bwd"]


# kernels behind each timed op (names as rocprofv3 prints them), for the roofline line and the PMC lookup
OP_KERNELS = {
    # (alternatives in a tuple: the one that ran.  Mode 5 = clipped SGD taken speculatively inside the reduce, mode 1 = plain
    # reduce followed by the clipped update of fg_row_update)
    "fg_reduce": [("fg_reduce_kernel<4, 8, 5, false>", "fg_reduce_kernel<4, 8, 1, false>"), ("fg_combine_kernel<4, 8, 5>", "fg_combine_kernel<4, 8, 1>")],
    "fg_row_update": [("fg_row_update_kernel<4, 5, true>", "fg_row_update_kernel<4, 2, true>", "fg_row_update_kernel<4, 2>")],
    "segreduce_fm": ["segreduce_chunk_kernel<4, 1, 1, true>", "segreduce_combine_kernel<4, 1, 1, true>"],
    "fm2_fwd": ["fm2_fwd_kernel<4, 1>"],
    "row_update": ["row_update_kernel<4, 1, 2, false>"],
}


def kernels_that_run(op, opt=None):
    """Names of the kernels behind ``op`` in this run: of alternatives, the speculative clipped-SGD variants when the
    optimizer takes that form (a clip bound, plain SGD, speculation not switched off), else the next one listed."""
    from recsys_amd import optim
    ks = OP_KERNELS.get(op)
    if ks is None:
        return None
    spec = (opt is not None and getattr(opt, "kind", None) == "sgd" and getattr(opt, "clip_norm", None) and getattr(opt, "speculative", False)
            and not optim._NO_SPECULATION)
    return [(k[0] if spec or len(k) == 1 else k[1]) if isinstance(k, tuple) else k for k in ks]


def unique_fraction(batches, dims, want_round=False):
    """rho = distinct table rows / occurrences, averaged over the synthetic batches.  ``want_round``:
    also the number of distinct rows over ALL resident batches (what a round of the timed region touches)."""
    offs = torch.tensor([0] + list(dims[:-1]), device=batches[0]["sparse"].device).cumsum(0)
    fr, seen = [], None
    for b in batches:
        flat = (b["sparse"] + offs[None, :]).reshape(-1)
        u = torch.unique(flat)
        fr.append(u.numel() / flat.numel())
        if want_round:
            seen = u if seen is None else torch.unique(torch.cat([seen, u]))
    rho = sum(fr) / len(fr)
    return (rho, int(seen.numel())) if want_round else rho


def algo_bytes(desc, op, rho=1.0):
    """Algorithmic HBM bytes of one launch (fp32 rows, 8-byte indices, N = 1 item per row):
    the per-sample figures are fwd F(8 + De) + 4, bwd 4 + F De + 8F, update rho F 2 De, with
    F fields, D columns, e = 4 bytes and rho the measured distinct-row fraction."""
    B, D = desc["batch_per_gpu"], desc["dim"]
    if desc["workload"] in ("fm_criteo", "fm_wide_d128", "fm_criteo_sharded", "deepfm_criteo", "deepfm_criteo_bf16", "dcnv2_criteo"):
        F = desc["fields"]
        if op == "fm2_fwd":  # index + row + bias entry per field, one logit out
            return B * (F * (8 + 4 * D + 4) + 4)
        if op == "fm2_bwd":  # logit grad in; indices + rows re-read; per-occurrence grad rows (+bias) out
            return B * (4 + F * (8 + 4 * D) + F * (4 * D + 4))
        if op == "segreduce":  # sorted positions + per-occurrence rows in; distinct rows read + written
            return B * F * (8 + 4 * D) + rho * B * F * 2 * 4 * D
        if op in ("segreduce_fm", "fg_reduce"):  # backward + the read-modify-write of the distinct rows, in one op
            return B * (4 + F * 4 * D + F * 8) + rho * B * F * 2 * 4 * D
        if op in ("row_update", "fg_row_update"):  # compact gradient in, row in, row out
            return rho * B * F * 3 * 4 * D
    return None


def embedding_bytes_per_sample(desc, rho):
    """Algorithmic HBM bytes of the embedding work of one sample over the whole step (fp32, plain SGD,
    N = 1, a D = 1 bias table beside the D-wide one): fwd F(8 + 4D + 4) + 4, bwd 4 + 4FD + 8F,
    update rho F 2 (4D + 4)."""
    F, D = desc["fields"], desc["dim"]
    return F * (8 + 4 * D + 4) + 4 + 4 + 4 * F * D + 8 * F + rho * F * 2 * (4 * D + 4)


# random 128-byte line accesses per second the card sustains (profiles/access_probe.py, uniform row reads of a 2 GiB
# D = 16 table, and of a 64 MiB one that the 256 MiB Infinity Cache holds): profiles/round2/access_probe.txt
LINE_CEILING_HBM, LINE_CEILING_CACHE = 38e9, 55e9


def line_accesses_per_step(desc, rho, speculative=True):
    """128-byte line accesses of one fm_criteo step (D = 16, rows in whole lines), the unit the memory system
    moves: ``random`` ones (one per row / record touched at an unpredictable address) and ``streaming`` ones
    (bytes of the sequentially read or written arrays / 128)."""
    B, F = desc["batch_per_gpu"], desc["fields"]
    n = B * F
    per_distinct = 3 if speculative else 6     # row read + row write + undo record | row read + compact write; compact, bias gradient, row read, row write
    random = n + n + rho * n * per_distinct    # forward row reads, backward record reads, distinct-row traffic
    stream_bytes = (2 * 8 * n            # index matrix: forward + grouping
                    + 8 * B + 128 * B    # labels, backward records written by the forward
                    + 2 * 4 * n          # transposed keys: written, read
                    + 2 * 4 * n          # packed words of the shared bins: written, read (upper bound)
                    + 2 * 8 * n)         # grouped rows + samples: written, read by the reduce
    return {"random": random, "streaming": stream_bytes / 128.0}


def workload_dtype(name):
    """Arithmetic type of the path: embedding work is fp32 everywhere; the bf16 workloads round the
    tower / cross GEMM operands to bf16 (fp32 accumulation, fp32 master weights)."""
    return "bf16" if name in ("dcnv2_criteo", "deepfm_criteo_bf16", "deepfm_wide_d128_sharded") else "f32"


def dense_flops_per_sample(name):
    """Algorithmic flops of the dense layers per sample, fwd + grad-x + grad-w (3 x 2 MACs)."""
    def mlp(dims):
        return sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    if name == "dcnv2_criteo":
        return 6.0 * (3 * 432 * 432 + mlp([432, 1024, 512, 256, 1]))
    if name in ("deepfm_criteo", "deepfm_criteo_bf16", "deepfm_criteo_sharded"):
        return 6.0 * mlp([432, 432, 216, 108, 1])
    if name == "deepfm_wide_d128_sharded":
        return 6.0 * mlp([3344, 3344, 1672, 836, 1])
    if name == "neumf_ml1m":  # per (user, item) pair: MLP 128-64-32, GMF 64-32, head 64-1; 5 pairs per sample
        return 6.0 * 5 * (mlp([128, 64, 32]) + 64 * 32 + 64)
    return None


def run_train_loop(name, device, clip):
    """Delivered rate of the reference's training loop on ML-1M-shaped data (cfg 1: VanillaMF, batch
    1024; cfg 2: NeuMF, batch 8192; D = 64): one epoch of ``recsys_amd.utils.train_epoch`` over
    ``ImplicitData``'s device-resident loader -- negative sampling for the epoch, the epoch shuffle,
    batch slicing, the copy of every batch into the recorded step's static tensors and the graph
    replay are all inside the timed region.  Epoch 0 (eager first steps, graph recording) is warm-up;
    epoch 1 is timed."""
    import torch
    from recsys_amd import ops
    from recsys_amd.dataset import ImplicitData, synthetic_ml1m_data
    from recsys_amd.models.matrix_factorization import VanillaMF
    from recsys_amd.models.neuralcf import NeuMF
    from recsys_amd.optim import FusedOptimizer
    from recsys_amd.utils import LoopGraph, train_epoch
    B = 1024 if name.startswith("mf_") else 8192
    data = ImplicitData(synthetic_ml1m_data(seed=0), device=device, train_batch_size=B, seed=1)
    torch.manual_seed(1234)
    with torch.device(device):
        model = (VanillaMF if name.startswith("mf_") else NeuMF)(data.n_users, data.n_items, 64)
    model.train()
    if name.endswith("_adam"):
        # the reference pipelines' own optimizer (recsys/pipelines/neuralcf/neumf.py:10: Adam(lr=1e-3) over all
        # parameters, dense table gradients, clip_grad_norm_ 1.0) turned into its fused, graph-replayable equivalent
        opt = FusedOptimizer.from_torch(torch.optim.Adam(model.parameters(), lr=1e-3), model, clip_norm=clip)
    else:
        opt = FusedOptimizer(model, kind="sgd", lr=0.05, clip_norm=clip)
    step = LoopGraph(model, opt)      # index checks are already deferred for the whole bench
    for epoch in range(2):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        dl = data.create_negative_sampled_train_dataloader()
        running = train_epoch(model, opt, dl, clip, step)
        torch.cuda.synchronize(device)
        el = time.perf_counter() - t0
    ops.raise_if_index_error()
    n = data.implicit_matrix.nnz
    return {"workload": name, "samples_per_s": n / el, "ms_per_step": 1e3 * el / len(dl), "steps": len(dl), "dtype": "f32",
            "epoch_mean_loss": float(running.item()) / len(dl),
            "optimizer": "Adam(lr=1e-3) on dense gradients of every parameter, as the reference pipelines build it" if name.endswith("_adam") else "fused sparse SGD",
            "what": "one epoch of train_epoch over ImplicitData's device loader, timed end to end: negative sampling and shuffle on "
                    "the device, then one graph launch per step (the recorded step ends with rx_epoch_advance: running loss, "
                    "copy of the next batch's slices into the static batch, cursor); the short last batch runs eagerly"}


def run_eval_loop(name, device):
    """The evaluation half of an epoch of the reference's loop (recsys/utils.py:67-81): list-wise forward of
    every test user over its 100 candidate items, top-10 on the device, NDCG / HR -- ``recsys_amd.utils.eval_loop``
    on ML-1M-sized data (6040 users), timed end to end including the metrics' host read."""
    import contextlib
    import io
    from recsys_amd.dataset import ImplicitData, synthetic_ml1m_data
    from recsys_amd.models.matrix_factorization import VanillaMF
    from recsys_amd.models.neuralcf import NeuMF
    from recsys_amd.utils import eval_loop
    data = ImplicitData(synthetic_ml1m_data(seed=0), device=device, seed=1)
    data.create_test_dataloader()
    torch.manual_seed(1234)
    with torch.device(device):
        model = (VanillaMF if name.startswith("mf_") else NeuMF)(data.n_users, data.n_items, 64)
    times = []
    for _ in range(6):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):     # eval_loop prints its metrics line like the reference
            ndcg, hr = eval_loop(model, data, 10)
        torch.cuda.synchronize(device)
        times.append(time.perf_counter() - t0)
    el = sorted(times[1:])[len(times[1:]) // 2]
    users = int(data.test_codes.shape[0])
    return {"workload": name, "users_per_s": users / el, "ms_per_eval": 1e3 * el, "users": users,
            "candidates_per_user": int(data.test_codes.shape[1]), "k": 10, "dtype": "f32", "ndcg": float(ndcg), "hr": float(hr),
            "what": "utils.eval_loop: N = 100 list-wise forwards over the test loader, rx_topk_codes, NDCG / HR; median of 5 after 1 warm-up"}


def kernel_source_sha():
    """sha256 over the HIP sources and the C header: identifies the kernels a PMC measurement was taken on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "recsys_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "recsys_amd", "csrc", "*.h"))
                    + [os.path.join(ROOT, "include", "recsys_hip.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(op):
    """HBM bytes per launch from committed rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE, summed
    over the op's kernels), or None when no measurement is on file — or when the file was taken on
    other kernel sources than the ones in the tree (its ``_source_sha`` stamp)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path) or op not in OP_KERNELS:
        return None
    table = json.load(open(path))
    if table.get("_source_sha") != kernel_source_sha():
        print("[bench] profiles/pmc_traffic.json was measured on other kernel sources: roofline.traffic omitted", file=sys.stderr)
        return None
    total = 0.0
    for k in OP_KERNELS[op]:
        names = [n for n in (k if isinstance(k, tuple) else (k,)) if n in table]   # alternatives: the one that ran
        if not names:
            return None
        total += table[names[0]]["traffic_bytes"]
    return total


# ----------------------------------------------------------------------------- CPU baseline
def cpu_baseline(name, desc, clip, budget_s=25.0, optimizer="sgd"):
    """The oracle's train step (dense gradients, clip_grad_norm_, torch optimizer) on the host cores.
    ``optimizer``: "sgd" matches the GPU step's update rule; "adam" is what the reference's pipelines
    build (Adam(lr=1e-3), e.g. recsys/pipelines/matrix_factorization/vanilla_mf.py:10)."""
    from oracle.fields import FieldOracle
    from oracle.models import OracleModel, init_state
    from recsys_amd import synthetic
    from recsys_amd.models import fields

    threads = torch.get_num_threads()
    B = desc["batch_per_gpu"]
    if name in ("fm_criteo", "deepfm_criteo", "deepfm_criteo_bf16", "dcnv2_criteo"):
        dims = synthetic.criteo_field_dims()
        cls = {"fm_criteo": "FieldFM", "deepfm_criteo": "FieldDeepFM", "deepfm_criteo_bf16": "FieldDeepFM",
               "dcnv2_criteo": "DCNv2"}[name]
        torch.manual_seed(1234)
        m = getattr(fields, cls)(dims, desc["dim"])
        oracle = FieldOracle(cls, m.state_dict(), dims, dropout=0.1)
        del m
        batches = [synthetic.criteo_batch(dims, B, seed=i, device="cpu") for i in range(2)]
    else:
        mname = "VanillaMF" if name == "mf_ml1m" else "NeuMF"
        st = init_state(mname, n_users=synthetic.ML1M["n_users"], n_items=synthetic.ML1M["n_items"],
                        embedding_dim=desc["dim"], seed=0)
        oracle = OracleModel(mname, st, n_users=synthetic.ML1M["n_users"], n_items=synthetic.ML1M["n_items"])
        batches = [synthetic.ml1m_batch(B, seed=i, device="cpu") for i in range(2)]
    opt = torch.optim.SGD(oracle.parameters(), lr=0.05) if optimizer == "sgd" else torch.optim.Adam(oracle.parameters(), lr=1e-3)

    def step(i):
        opt.zero_grad()
        loss = oracle.loss(batches[i % 2])
        loss.backward()
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(oracle.parameters(), clip)
        opt.step()
        return loss.item()

    step(0)  # warm-up
    t0, n = time.perf_counter(), 0
    while True:
        step(n)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 50:
            break
    return {"value": B * n / el, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{n} steps of the same workload (batch {B}) after 1 warm-up step, dense gradients + "
                      f"clip_grad_norm_ + torch.optim.{'SGD' if optimizer == 'sgd' else 'Adam(lr=1e-3)'}, {el:.1f} s"}


def extra_entry(name, device, clip, run):
    """One entry of ``other_workloads``: a further single-GPU workload timed briefly."""
    if name.endswith("_eval_loop"):
        return run_eval_loop(name, device)
    if name.endswith("_train_loop") or name.endswith("_train_loop_adam"):
        return run_train_loop(name, device, clip)
    # the 64 GB-table workloads need a longer warm-up (allocator growth, first touches)
    n2, wu2 = (96, 24) if name in ("fm_wide_d128", "fm_criteo_sharded") else (30, 5)
    if name == "deepfm_wide_d128_sharded":
        n2, wu2 = 12, 3        # 20-30 ms per step; the 24 set-up steps of run() come on top
    if name in ("mf_ml1m", "neumf_ml1m"):
        n2, wu2 = 200, 20      # steps of 0.1-0.4 ms: a 30-step region is a few ms and one hiccup doubles it
    w2, el2, _, _ = run(name, n2, wu2)
    entry = {"workload": name, "samples_per_s": w2["B"] * n2 / el2, "ms_per_step": 1e3 * el2 / n2,
             "dtype": workload_dtype(name)}
    if name in ("fm_wide_d128", "fm_criteo_sharded"):
        r2 = unique_fraction(w2["batches"], w2["dims"])
        b2 = embedding_bytes_per_sample(w2["desc"], r2)
        entry["embedding_hbm_frac"] = b2 * entry["samples_per_s"] / 1e9 / HBM_PEAK_GBS
        entry["distinct_row_fraction"] = r2
    if name == "deepfm_wide_d128_sharded":
        entry.update(rows=w2["desc"]["rows"], dim=w2["desc"]["dim"],
                     role="the 1 B-row x 128 DeepFM as named, one rank's share (125 M rows, tower 3344-3344-1672-836-1)")
    if name == "fm_criteo_sharded":
        # what `--gpus N` runs for N > 1, here with one rank: the base of the weak-scaling series.  At N > 1 the
        # step is launched eagerly (recorded collectives were never rehearsed on several GPUs), at one rank it
        # replays a graph: the same one-rank step launched eagerly is timed too, so that the series has a base
        # in its own launch mode
        entry.update(rows=w2["desc"]["rows"], dim=w2["desc"]["dim"], role="one-rank base of the multi-GPU series",
                     launch="one HIP graph per step")
        del w2
        torch.cuda.empty_cache()
        w3, el3, _, _ = run(name, n2, wu2, use_graph=False)
        entry["eager_launch"] = {"ms_per_step": 1e3 * el3 / n2, "samples_per_s": w3["B"] * n2 / el3,
                                 "note": "the launch mode of --gpus N > 1: compare the N-GPU line with this figure"}
        del w3
    fl = dense_flops_per_sample(name)
    if fl:  # whole-step rate of the tower / cross GEMM work (fwd + both gradients)
        entry["dense_tflops_per_s"] = fl * entry["samples_per_s"] / 1e12
    return entry


def extra_in_child(name, args, limit_s=420):
    """``extra_entry`` in a child process (this file with --child-entry); a child that fails leaves an
    entry that says so.  The parent has released its tables before this is called."""
    import subprocess
    cmd = [sys.executable, "-u", os.path.abspath(__file__), "--workload", name, "--child-entry", "--extra", "none",
           "--no-cpu-baseline"] + (["--no-clip"] if args.no_clip else [])
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=limit_s)
    except subprocess.TimeoutExpired:
        return {"workload": name, "error": f"no result within {limit_s} s"}
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"workload": name, "error": f"child process ended with code {p.returncode}", "stderr_tail": p.stderr[-300:]}
    return json.loads(lines[-1])


# ----------------------------------------------------------------------------- main
def main():
    # stdout carries the result line and nothing else: libraries that write to file descriptor 1 on their own
    # (RCCL's version banner at communicator set-up, gloo's connection messages) are sent to stderr
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="fm_criteo")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the median one is reported")
    ap.add_argument("--no-clip", action="store_true", help="skip the global grad-norm clip (one-phase fused update)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying a HIP graph")
    ap.add_argument("--op-profile", action="store_true", help="print average HIP-event time of every op to stderr")
    ap.add_argument("--extras-in-process", action="store_true", help="time the --extra workloads in this process instead of one child process each")
    ap.add_argument("--child-entry", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--extra", default="mf_ml1m,mf_ml1m_train_loop,mf_ml1m_train_loop_adam,neumf_ml1m,neumf_ml1m_train_loop,neumf_ml1m_train_loop_adam,neumf_ml1m_eval_loop,deepfm_criteo,deepfm_criteo_bf16,dcnv2_criteo,fm_wide_d128,fm_criteo_sharded,deepfm_wide_d128_sharded",
                    help="comma list of further single-GPU workloads to time briefly and append (\"none\" to skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("RX_BENCH_WATCHDOG"):   # seconds; a stuck run prints every thread's stack and exits
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["RX_BENCH_WATCHDOG"]), exit=True)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("RX_BENCH_SHARE_GPU") == "1":   # rehearsal on a one-GPU machine: every rank opens card 0
        local = 0
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    force_dist = os.environ.get("RX_BENCH_FORCE_DIST") == "1"   # rehearse the multi-rank code path with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:     # RX_BENCH_FORCE_DIST without a launcher: a group of one
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_PORT", "29533")):
                os.environ.setdefault(k, v)
        backend = os.environ.get("RX_BENCH_BACKEND", "nccl")   # "gloo" only for the shared-card rehearsal
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
            # gloo's device all-to-all stalls now and then with tens of MB in flight between two
            # processes on one card; the rehearsal sends host copies instead
            inner = dist.all_to_all_single

            def staged(out, inp, *rest, **kw):
                o = torch.empty(out.shape, dtype=out.dtype)
                inner(o, inp.cpu(), *rest, **kw)
                out.copy_(o)
            dist.all_to_all_single = staged

    from recsys_amd import ops
    ops.set_index_check("deferred")
    clip = None if args.no_clip else float(os.environ.get("RX_BENCH_CLIP_NORM", 1.0))   # (the reference clips at 1.0: recsys/utils.py:101)

    def run(name, steps, warmup, timed_op=None, use_graph=True, repeats=1, check=False):
        """Builds the workload, runs `warmup` untimed steps, then times `repeats` regions of EXACTLY `steps`
        steps each (barrier + synchronize on both sides, maximum over the ranks); returns the median
        region's time, and the list of all of them in w["region_s"]."""
        w = build_workload(name, device, rank, world, clip, use_graph)
        # the timed region starts on resident batch 0, where a whole-round graph may begin (any slot
        # can go first: all were prepared when they were registered)
        first = (-warmup) % len(w["batches"])
        if name in ("fm_criteo_sharded", "deepfm_criteo_sharded", "deepfm_wide_d128_sharded", "fm_wide_d128") and warmup < 24:
            # part of set-up, whatever --warmup says: the 64 GB tables and the exchange buffers are first
            # touched, and the allocator grows, over the first two dozen steps
            run_steps(w, 0, 24 - warmup)
            w.pop("prepared", None)
        run_steps(w, first, warmup)
        # the bf16-storage towers run the grad-W product under another op name
        timer = OpTimer(([timed_op] + (["linear16_bwd_w"] if timed_op == "linear_bwd_w" else [])) if timed_op else [])
        pos, regions = first + warmup, []
        gs, K = w.get("graphed"), len(w["batches"])
        for _ in range(repeats):
            # (between repeats: untimed steps up to the next slot where a multi-step graph starts)
            while regions and gs is not None and gs.segments and not gs.segment_at(pos % K):
                run_steps(w, pos, 1)
                pos += 1
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            with timer:
                loss = run_steps(w, pos, steps)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
                torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([el], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            regions.append(el)
            pos += steps
        if check:
            w["self_check"], pos = self_check(w, pos)
        ops.raise_if_index_error()
        w["region_s"] = regions
        el = sorted(regions)[len(regions) // 2]
        ops_ms = timer.summary()
        ran = next((n for n in reversed(timer.names) if n in timer.last_call), None) if timed_op else None
        if ran is not None:
            ops_ms[timed_op] = back_to_back_ms(ran, timer.last_call[ran])
            if timed_op == "linear_bwd_w":      # dW[N, K] = dY[M, N]^T X[M, K]: the shape prices the launch
                dy_, x_ = timer.last_call[ran][0][:2]
                ops_ms["_gemm_shape"] = (int(dy_.shape[0]), int(dy_.shape[1]), int(x_.shape[1]))
                ops_ms["_gemm_op"] = ran
        return w, el, ops_ms, float(loss.item())

    if args.child_entry:       # one entry of other_workloads, printed for the parent process
        print(json.dumps(extra_entry(args.workload, device, clip, run)), file=result_out, flush=True)
        return
    if args.op_profile and rank == 0:
        w = build_workload(args.workload, device, rank, world, clip, use_graph=False)
        for i in range(5):
            one_step(w, i)
        with OpTimer(ALL_OPS) as t:
            for i in range(20):
                one_step(w, i)
        for n, (ms, cnt) in sorted(t.summary().items(), key=lambda kv: -kv[1][0] * kv[1][1]):
            print(f"[op-profile] {n:22s} {ms * 1e3:9.1f} us x {cnt / 20:.1f}/step", file=sys.stderr)
        del w

    fg_on = os.environ.get("RX_NO_FIELD_GROUPS") != "1"   # the unsharded field models take the field-grouped backward
    fg_on = fg_on and int(os.environ.get("RX_BENCH_DIM", 16)) <= int(os.environ.get("RX_FG_MAX_DIM", 32))   # wide rows take the general path
    dominant = {"fm_criteo": "fg_reduce" if fg_on else "segreduce_fm", "fm_wide_d128": "segreduce_fm",
                "fm_criteo_sharded": "segreduce_fm",
                "deepfm_criteo_sharded": "linear_bwd_w", "deepfm_wide_d128_sharded": "linear_bwd_w", "deepfm_criteo": "linear_bwd_w",
                "deepfm_criteo_bf16": "linear_bwd_w",
                "dcnv2_criteo": "linear_bwd_w", "mf_ml1m": "mf_score_bwd", "neumf_ml1m": "linear_bwd_w"}[args.workload]
    use_graph = not args.no_graph
    if (world > 1 or force_dist) and args.workload == "fm_criteo":
        args.workload = "fm_criteo_sharded"
    if args.workload in ("fm_criteo_sharded", "deepfm_criteo_sharded", "deepfm_wide_d128_sharded"):
        # graphs with the exchanges inside: at one rank by default, at more ranks on request (graph_sharded)
        want = os.environ.get("RX_BENCH_SHARDED_GRAPH")
        use_graph = use_graph and want != "0" and (world == 1 or want == "1")
    w, el, ops_ms, last_loss = run(args.workload, args.steps, args.warmup, None if use_graph else dominant, use_graph,
                                   repeats=args.repeats, check=True)
    if use_graph:
        # the dominant kernel is timed with HIP events on the launch stream after a short eager pass of
        # the same steps (events cannot bracket a single node of a replayed graph)
        _, _, ops_ms, _ = run(args.workload, min(args.steps, 30), 5, dominant, use_graph=False)
    B = w["B"]
    value = B * world * args.steps / el
    out = {
        "metric": "train samples/sec (fwd+bwd+update)", "value": value, "unit": "samples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
        "timed_regions": {"count": len(w["region_s"]), "steps_each": args.steps, "reported": "median",
                          "ms_per_step": [round(1e3 * x / args.steps, 5) for x in w["region_s"]],
                          "min": 1e3 * min(w["region_s"]) / args.steps, "max": 1e3 * max(w["region_s"]) / args.steps},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": workload_dtype(args.workload), "data": "synthetic",
        "config": dict(w["desc"], global_batch=B * world, parallelism=f"dp{world}" if world > 1 else "single"),
        "final_loss": last_loss, "self_check": w.get("self_check"),
        "grad_norm_last_step": (float(w["opt"].last_total_norm.item()) if getattr(w["opt"], "last_total_norm", None) is not None else None),
    }
    roof = None
    rho, round_rows = 1.0, None
    if "fields" in w["desc"]:
        from recsys_amd import synthetic
        rho, round_rows = unique_fraction(w["batches"], w.get("dims") or synthetic.criteo_field_dims(), want_round=True)
    if dominant in ops_ms:
        ms, cnt = ops_ms[dominant]
        ab = algo_bytes(w["desc"], dominant, rho)
        if ab:
            ach = ab / (ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dominant,
                    "kernels": kernels_that_run(dominant, w.get("opt")) if w["desc"].get("dim") == 16 else None, "achieved": ach,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": measured_traffic(dominant) if args.workload == "fm_criteo" else None, "avg_us": ms * 1e3, "launches_timed": cnt,
                    "algorithmic_bytes_per_launch": ab, "distinct_row_fraction": rho}
    if roof is None and dominant == "linear_bwd_w" and "_gemm_shape" in ops_ms and dominant in ops_ms:
        ms, cnt = ops_ms[dominant]
        M_, N_, K_ = ops_ms["_gemm_shape"]
        bf = workload_dtype(args.workload) == "bf16"
        peak = 2500.0 if bf else 157.0        # dense MFMA peaks, TFLOP/s
        ach = 2.0 * M_ * N_ * K_ / (ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": f"{ops_ms.get('_gemm_op', 'linear_bwd_w')} (the grad-W launched last in the step, i.e. the first layer's, with its slab and bias sums)",
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None, "avg_us": ms * 1e3,
                "launches_timed": cnt, "algorithmic_flops_per_launch": 2.0 * M_ * N_ * K_, "shape_M_N_K": [M_, N_, K_]}
    out["roofline"] = roof
    if args.workload in ("fm_criteo", "fm_wide_d128", "fm_criteo_sharded"):
        bps = embedding_bytes_per_sample(w["desc"], rho)
        out["embedding_hbm"] = {"algorithmic_bytes_per_sample": bps, "achieved": bps * value / world / 1e9, "unit": "GB/s per GPU",
                                "peak": HBM_PEAK_GBS, "frac": bps * value / world / 1e9 / HBM_PEAK_GBS,
                                "note": "whole step (fwd + bwd + update), all kernels, at the measured distinct-row fraction",
                                "distinct_row_fraction": rho, "resident_batches": len(w["batches"]),
                                "distinct_rows_per_round": round_rows,
                                "table_bytes_touched_per_round": None if round_rows is None else round_rows * (4 * w["desc"]["dim"] + 4),
                                "infinity_cache_bytes": 256 * 1024 * 1024}
    if args.workload == "fm_criteo" and w["desc"].get("dim") == 16:
        spec = os.environ.get("RX_NO_SPECULATIVE_SGD") != "1" and clip is not None
        la = line_accesses_per_step(w["desc"], rho, spec)
        tot = la["random"] + la["streaming"]
        out["roofline_lines"] = {
            "lines_per_step": tot, "random_lines_per_step": la["random"], "streaming_lines_per_step": la["streaming"],
            "achieved_Glines_s": tot / (el / args.steps) / 1e9,
            "ceiling_Glines_s": {"random_from_hbm": LINE_CEILING_HBM / 1e9, "random_from_infinity_cache": LINE_CEILING_CACHE / 1e9},
            "frac_of_hbm_random_ceiling": tot / (el / args.steps) / LINE_CEILING_HBM,
            "note": "a D = 16 row is half of a 128-byte line: by lines moved, not by bytes asked for, is how this step is bound "
                    "(profiles/round3/fetch_size_calibration.txt: 2 x FETCH_SIZE equals the lines moved for 64-byte and 128-byte rows alike)"}
    if world > 1:
        # what a weak-scaling efficiency of this line is to be computed against: the SAME sharded workload at
        # one rank, which the --gpus 1 line reports under other_workloads (its main line is the unsharded
        # D = 16 configuration, another workload)
        out["scaling_base"] = {"workload": args.workload, "where": "other_workloads[workload == '%s'] of the --gpus 1 line" % args.workload,
                               "launch": "graph replay" if use_graph else "eager",
                               "note": "value / (n_gpus x that entry's samples_per_s); this line was launched "
                                       + ("as graph replays: compare with the entry's own figure" if use_graph else
                                          "eagerly: compare with the entry's eager_launch figure")
                                       + "; the --gpus 1 main line runs fm_criteo (D = 16, unsharded)"}
    cap_slots = getattr(w["model"], "capacity", None)
    if world > 1 and cap_slots:
        # third bound of the sharded step: bytes a rank puts on its xGMI links per step (slot ids out,
        # rows + bias entries back, gradients of both out), the share addressed to itself excluded
        D = w["desc"]["dim"]
        sent = (world - 1) * cap_slots * (8 + 2 * (4 * D + 4))
        out["xgmi"] = {"bytes_sent_per_gpu_per_step": sent, "achieved": sent / (el / args.steps) / 1e9, "unit": "GB/s per GPU",
                       "peak": 1070.0, "frac": sent / (el / args.steps) / 1e9 / 1070.0,
                       "note": "egress per GPU against 7 links x 153 GB/s; exchanges are not overlapped with compute yet"}
    del w
    torch.cuda.empty_cache()

    extras = [x for x in args.extra.split(",") if x and x != "none" and x != args.workload]
    if extras and rank == 0 and world == 1:
        # every further workload in a process of its own: a crash or a hang there costs that entry, not the line
        out["other_workloads"] = [extra_entry(name, device, clip, run) if args.extras_in_process else extra_in_child(name, args)
                                  for name in extras]

    if rank == 0 and world == 1 and args.workload == "fm_criteo" and os.environ.get("RX_BENCH_BATCHES") is None:
        # the same workload cycling over 10 resident batches: the rows a round
        # touches then fit the 256 MiB Infinity Cache, so part of the table traffic never reaches HBM
        os.environ["RX_BENCH_BATCHES"] = "10"
        w10, el10, _, _ = run(args.workload, args.steps, args.warmup, None, use_graph, repeats=3)
        del os.environ["RX_BENCH_BATCHES"]
        r10, rows10 = unique_fraction(w10["batches"], w10["dims"], want_round=True)
        b10 = embedding_bytes_per_sample(w10["desc"], r10)
        out["embedding_hbm"]["with_10_resident_batches"] = {
            "ms_per_step": 1e3 * el10 / args.steps, "frac": b10 * w10["B"] * args.steps / el10 / 1e9 / HBM_PEAK_GBS,
            "table_bytes_touched_per_round": rows10 * (4 * w10["desc"]["dim"] + 4)}
        del w10
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload, out["config"], clip)
        out["cpu_baseline_adam"] = cpu_baseline(args.workload, out["config"], clip, budget_s=12.0, optimizer="adam")
    if rank == 0:
        print(json.dumps(out), file=result_out, flush=True)
    if str(out.get("self_check", "")).startswith("FAILED"):
        raise SystemExit("bench.py: " + out["self_check"])
    if dist is not None:
        # recorded collectives must be gone before the group is torn down (the teardown blocks otherwise)
        import gc
        gc.collect()
        torch.cuda.synchronize(device)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
