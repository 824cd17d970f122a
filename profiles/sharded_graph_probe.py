#!/usr/bin/env python3
"""One rank, RCCL group, exchanges forced on: record the fixed-capacity sharded step (FM, or the
DeepFM with its tower all-reduce: MODEL=deepfm) as HIP graphs (graph.ShardedGraphStep) and compare
the replays with eager steps on the same batch sequence -- weights bit for bit.  Prints both
timings.  ROWS / DIM / B size the run (defaults: the bench's batch on a 2 M-row table)."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from recsys_amd import ops, synthetic  # noqa: E402
from recsys_amd.graph import ShardedGraphStep  # noqa: E402
from recsys_amd.sharded import ShardedFieldDeepFM, ShardedFieldFM  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
rows = int(os.environ.get("ROWS", 2_000_000))
D = int(os.environ.get("DIM", 128))
B = int(os.environ.get("B", 65536))
deep = os.environ.get("MODEL", "fm") == "deepfm"
K, STEPS = 3, 12
dims = synthetic.criteo_field_dims(total_rows=rows)
batches = [synthetic.criteo_batch(dims, B, seed=i, device=dev) for i in range(K)]
ops.set_index_check("deferred")


def build():
    torch.manual_seed(1)
    m = ShardedFieldDeepFM(dims, D, dev, dropout=0.0) if deep else ShardedFieldFM(dims, D, dev)
    m.set_rule("sgd", 0.05, clip_norm=1.0)
    m.feature.force_exchange = m.bias.force_exchange = True
    m.calibrate(batches, slack=1.1)
    m.train()

    def step(b):
        if deep:
            m.zero_dense_grads()
        loss = m.loss(b)
        loss.backward()
        if deep:
            m.allreduce_dense_grads()
            m.dense_sgd_step(0.05, m.finish_step(m.dense_grads()))
        else:
            m.finish_step()
        return loss.detach()
    return m, step


def timed(fn, first, count):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(first, first + count):
        out = fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / count * 1e3, out


a, step_a = build()
for i in range(3):
    step_a(batches[0])                       # the graph side's warm-up steps, mirrored
ms_e, la = timed(lambda i: step_a(batches[i % K]), 0, STEPS)
print("eager ms/step", ms_e, flush=True)

b, step_b = build()
gs = ShardedGraphStep(step_b, batches, dev, warmup=3, model=b if os.environ.get("AHEAD", "1") == "1" else None)
print("index work one batch ahead:", gs.ahead, flush=True)
print("recorded", flush=True)
ms_g, lb = timed(lambda i: gs(i % K), 0, STEPS)
print("graph ms/step", ms_g, flush=True)
ops.raise_if_index_error()
same = torch.equal(a.feature.weight, b.feature.weight) and torch.equal(a.bias.weight, b.bias.weight)
if deep:
    same = same and all(torch.equal(x, y) for x, y in zip(a.deep_mlp.state_dict().values(), b.deep_mlp.state_dict().values()))
print("loss", float(la), float(lb))
print("bitwise equal", same and float(la) == float(lb), flush=True)
gs.release()
del gs
torch.cuda.synchronize()
dist.destroy_process_group()
print("group destroyed", flush=True)
