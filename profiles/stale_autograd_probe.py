#!/usr/bin/env python3
"""A host crash of the graph path, reproduced: recording a train step while the autograd graph of an earlier
EAGER step is still alive.

Two eager steps on the default stream; the last loss stays in a variable, so its autograd graph -- and with it
the gradient-accumulation node of every parameter that gets a dense gradient, bound to the default stream --
survives.  The recording's backward pass then runs those nodes on the default stream from inside the capture.
With dense table gradients (kind=dense_adam) this run ends in a segmentation fault inside
hipStreamEndCapture (torch/cuda/graphs.py capture_end); with a sparse rule on the tables (kind=sgd: no
parameter of VanillaMF gets a dense gradient) it records and replays.  graph.refuse_live_autograd_graphs turns
the first case into a RuntimeError before anything is recorded; RX_ALLOW_LIVE_AUTOGRAD=1 (set below) switches
that check off for this probe.  Output of the run that found it: profiles/round3/stale_autograd_probe.txt.

    python profiles/stale_autograd_probe.py dense_adam|sgd
"""
import faulthandler
import os
import sys

import torch

os.environ["RX_ALLOW_LIVE_AUTOGRAD"] = "1"
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_amd import ops, synthetic  # noqa: E402
from recsys_amd.graph import GraphedTrainStep  # noqa: E402
from recsys_amd.models.matrix_factorization import VanillaMF  # noqa: E402
from recsys_amd.optim import FusedOptimizer  # noqa: E402

d = synthetic.ML1M
batches = [synthetic.ml1m_batch(256, seed=20 + i, device="cuda") for i in range(3)]
torch.manual_seed(5)
with torch.device("cuda"):
    m = VanillaMF(d["n_users"], d["n_items"], 16)
m.train()
kind = sys.argv[1] if len(sys.argv) > 1 else "dense_adam"
opt = FusedOptimizer(m, kind=kind, lr=1e-2, clip_norm=1.0)
ops.set_index_check("deferred")
for b in batches[:2]:
    opt.zero_grad(); l = m.loss(b); l.backward(); opt.step()
print("eager done; the last loss (with its autograd graph) is still alive", flush=True)
gs = GraphedTrainStep(m, opt, warmup=0)
slot = gs.add_batch({k: v.clone() for k, v in batches[2].items()})
print("recorded", flush=True)
torch.cuda.synchronize()
for i in range(3):
    gs(slot)
    torch.cuda.synchronize()
    print("replayed", i, flush=True)
