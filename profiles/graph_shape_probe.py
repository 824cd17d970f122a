#!/usr/bin/env python3
"""Shape of the recorded NeuMF loop graphs, two-branch against three-branch, WITHOUT replaying them.

A replay of the three-branch recording (the next batch's index sorts on a forked stream of their own,
beside the GMF half's branch and the MLP chain) ended in a host segmentation fault inside hipGraphLaunch
now and then; the two-branch recording (those sorts behind the GMF half on its stream) has not.  This
probe records both forms of the eight-step pair graph of utils.LoopGraph (cfg 2: NeuMF, batch 8192,
D = 64) keeping the hipGraph_t (keep_graph=True) and prints, from the runtime's own graph queries, per
recording: nodes by kind, edges, roots, leaves, the widest level of a longest-path layering (an upper
bound on the streams a launch needs at once) and the longest chain; plus the number of HIP streams
the process has created by then.  Nothing is replayed.

    python profiles/graph_shape_probe.py [outdir]
"""
import collections
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RX_GRAPH_MAX_WIDTH"] = "0"      # graph.py refuses three-wide recordings otherwise; nothing is replayed here


_KINDS = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record"}


def graph_stats(raw_graph: int):
    """Nodes / edges of a hipGraph_t through the runtime's own queries (hipGraphGetNodes, hipGraphGetEdges,
    hipGraphNodeGetType)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    g = C.c_void_p(raw_graph)
    n = C.c_size_t(0)
    assert hip.hipGraphGetNodes(g, None, C.byref(n)) == 0
    nodes = (C.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(g, nodes, C.byref(n)) == 0
    ne = C.c_size_t(0)
    assert hip.hipGraphGetEdges(g, None, None, C.byref(ne)) == 0
    src, dst = (C.c_void_p * ne.value)(), (C.c_void_p * ne.value)()
    assert hip.hipGraphGetEdges(g, src, dst, C.byref(ne)) == 0
    kinds = collections.Counter()
    for i in range(n.value):
        t = C.c_int(-1)
        assert hip.hipGraphNodeGetType(C.c_void_p(nodes[i]), C.byref(t)) == 0
        kinds[_KINDS.get(t.value, str(t.value))] += 1
    names = [nodes[i] for i in range(n.value)]
    edges = [(src[i], dst[i]) for i in range(ne.value)]
    succ, pred = collections.defaultdict(list), collections.defaultdict(list)
    for a, b in edges:
        succ[a].append(b)
        pred[b].append(a)
    level, done, indeg = {}, 0, {x: len(pred[x]) for x in names}
    todo = [x for x in names if indeg[x] == 0]
    for x in todo:
        level[x] = 0
    while todo:
        x = todo.pop()
        done += 1
        for s2 in succ[x]:
            level[s2] = max(level.get(s2, 0), level[x] + 1)
            indeg[s2] -= 1
            if indeg[s2] == 0:
                todo.append(s2)
    per_level = collections.Counter(level.values())
    # minimum number of chains that cover the DAG = its width (Dilworth) needs a matching; a cheap, exact-enough
    # figure for these near-series-parallel recordings is the number of nodes alive at once in level order
    return dict(nodes=len(names), edges=len(edges), roots=sum(1 for x in names if not pred[x]),
                leaves=sum(1 for x in names if not succ[x]), widest_level=max(per_level.values(), default=0),
                levels_with_3_or_more=sum(1 for v in per_level.values() if v >= 3),
                longest_chain=max(level.values(), default=-1) + 1,
                max_fan_in=max((len(pred[x]) for x in names), default=0),
                max_fan_out=max((len(succ[x]) for x in names), default=0), kinds=dict(kinds), acyclic=done == len(names))


def record(own_fork: bool, outdir: str):
    from recsys_amd import ops
    from recsys_amd.dataset import ImplicitData, synthetic_ml1m_data
    from recsys_amd.graph import GraphedTrainStep
    from recsys_amd.models.neuralcf import NeuMF
    from recsys_amd.optim import FusedOptimizer
    from recsys_amd.utils import train_step
    os.environ["RX_GRAPH_OWN_FORK"] = "1" if own_fork else "0"
    dev = torch.device("cuda", 0)
    data = ImplicitData(synthetic_ml1m_data(seed=0), device=dev, train_batch_size=8192, seed=1)
    torch.manual_seed(1234)
    with torch.device(dev):
        model = NeuMF(data.n_users, data.n_items, 64)
    model.train()
    opt = FusedOptimizer(model, kind="sgd", lr=0.05, clip_norm=1.0)
    dl = data.create_negative_sampled_train_dataloader()
    it = iter(dl)
    for _ in range(2):                      # eager steps: every lazily sized buffer at its final size
        train_step(model, opt, next(it), 1.0)
    pair = [{k: v.clone() for k, v in next(it).items()} for _ in range(2)]
    made = []
    real = torch.cuda.CUDAGraph

    def debug_graph(*a, **k):
        k["keep_graph"] = True
        g = real(*a, **k)
        made.append(g)
        return g

    torch.cuda.CUDAGraph = debug_graph
    try:
        gs = GraphedTrainStep(model, opt, warmup=0)
        gs.add_pair(pair, [lambda loss: None, lambda loss: None], unroll=8)
    finally:
        torch.cuda.CUDAGraph = real
    torch.cuda.synchronize()
    out = []
    for i, g in enumerate(made):
        what = f"neumf_pair_{'three' if own_fork else 'two'}_branch_graph{i}" + ("_8_steps" if i == 2 else "_1_step")
        out.append((what, graph_stats(g.raw_cuda_graph())))
    del gs, made
    return out


def main():
    outdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "graph_shape")
    os.makedirs(outdir, exist_ok=True)
    from recsys_amd import ops
    ops.set_index_check("deferred")
    created = [0]
    real_stream = torch.cuda.Stream

    class CountingStream(real_stream):
        def __new__(cls, *a, **k):
            if "stream_id" not in k and "stream_ptr" not in k:      # a new HIP stream, not a wrapper of an existing one
                created[0] += 1
            return super().__new__(cls, *a, **k)

    torch.cuda.Stream = CountingStream
    for own_fork in (False, True):
        before = created[0]
        for path, st in record(own_fork, outdir):
            print(("three-branch" if own_fork else "two-branch  "), os.path.basename(path), st, flush=True)
        print(f"  HIP streams created through torch.cuda.Stream() for this recording set: {created[0] - before}", flush=True)
    print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(unset: HIP's default, 4)"))


if __name__ == "__main__":
    main()
