#!/usr/bin/env python3
"""Where do the two chains of the pipelined graph sit in time, without a profiler attached?

Device wall-clock stamps (ops.wall_clock, 10 ns ticks) are captured into the graph around the
main chain (forward, loss, backward, update of batch i) and around the side chain (index sort of
batch i+1); after a run of replays the stamps of the last replay of every slot are printed
relative to the start of that replay's main chain.

Usage: overlap_probe.py [workload] [replays]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from recsys_amd import ops  # noqa: E402
from recsys_amd.graph import GraphedTrainStep  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "fm_criteo"
    replays = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ops.set_index_check("deferred")
    w = bench.build_workload(name, dev, 0, 1, 1.0, use_graph=False)
    model, opt, batches = w["model"], w["opt"], w["batches"]
    K = len(batches)
    stamps = torch.zeros(K, 4, dtype=torch.int64, device=dev)
    state = {"slot": None}

    gs = GraphedTrainStep(model, opt, pipeline=True)
    real_prepare, real_eager = model.prepare, gs._eager

    def prepare(b, out=None):
        if out is None or state["slot"] is None:
            return real_prepare(b, out=out)
        ops.wall_clock(stamps[state["slot"], 2:3])
        r = real_prepare(b, out=out)
        ops.wall_clock(stamps[state["slot"], 3:4])
        return r

    def eager(b):
        if state["slot"] is None:
            return real_eager(b)
        ops.wall_clock(stamps[state["slot"], 0:1])
        r = real_eager(b)
        ops.wall_clock(stamps[state["slot"], 1:2])
        return r

    model.prepare, gs._eager = prepare, eager
    # add_batches records slot i's graph in order; count the recordings through the warm-up flag
    real_graph = torch.cuda.graph
    counter = {"i": -1}

    class counting_graph(real_graph):
        def __enter__(self):
            counter["i"] += 1
            state["slot"] = counter["i"]
            return super().__enter__()

        def __exit__(self, *a):
            state["slot"] = None
            return super().__exit__(*a)

    torch.cuda.graph = counting_graph
    try:
        gs.add_batches(batches)
    finally:
        torch.cuda.graph = real_graph
    for i in range(20):
        gs(i % K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(replays):
        gs(i % K)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{name}: {1e6 * el / replays:.1f} us per replay")
    s = stamps.cpu().double() / 100.0   # us
    order = sorted(range(K), key=lambda k: s[k, 0].item())
    print("slot  main[start end]   side[start end]   next main start   (us, relative to this replay's main start)")
    for n, k in enumerate(order):
        base = s[k, 0].item()
        nxt = s[order[n + 1], 0].item() - base if n + 1 < K else float("nan")
        print(f"{k:4d}  {0.0:7.1f} {s[k, 1].item() - base:7.1f}   {s[k, 2].item() - base:7.1f} {s[k, 3].item() - base:7.1f}   {nxt:7.1f}")
    ops.raise_if_index_error()


if __name__ == "__main__":
    main()
