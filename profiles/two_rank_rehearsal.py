"""Several ranks on ONE card with the real kernels at bench-like sizes, collectives over gloo: a
rehearsal of the sharded step for machines with a single GPU.  Prints a line per phase (with a
device synchronisation in between when --sync is given) so that a stall names its phase.
    python profiles/two_rank_rehearsal.py --world 2 --rows 2000000 --dim 16 --steps 6 [--sync] [--prepare]"""
import argparse
import faulthandler
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, a):
    faulthandler.dump_traceback_later(a.watchdog, exit=True)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from recsys_amd import ops, synthetic
    from recsys_amd.sharded import ShardedFieldFM
    ops.set_index_check("deferred")
    t0 = time.perf_counter()

    def say(msg):
        if a.sync:
            torch.cuda.synchronize()
        print(f"[rank {rank} {time.perf_counter() - t0:7.2f}s] {msg}", flush=True)

    if a.trace:
        native = dist.all_to_all_single

        def traced(out, inp, *rest, **kw):
            torch.cuda.synchronize()
            print(f"[rank {rank}] all-to-all in: {tuple(inp.shape)} {inp.dtype}", flush=True)
            native(out, inp, *rest, **kw)
            print(f"[rank {rank}] all-to-all returned", flush=True)
            torch.cuda.synchronize()
            print(f"[rank {rank}] all-to-all out on the device", flush=True)
        dist.all_to_all_single = traced

    if a.stage_host:
        inner = dist.all_to_all_single

        def staged(out, inp, *rest, **kw):
            o = torch.empty(out.shape, dtype=out.dtype)
            inner(o, inp.cpu(), *rest, **kw)
            out.copy_(o)
        dist.all_to_all_single = staged

    dims = synthetic.criteo_field_dims(total_rows=a.rows * world)
    model = ShardedFieldFM(dims, a.dim, torch.device("cuda", 0))
    model.set_rule("sgd", 0.05, clip_norm=1.0)
    batches = [synthetic.criteo_batch(dims, a.batch, seed=1000 * rank + i, device="cuda") for i in range(3)]
    say("batches built")
    cap = model.calibrate(batches)
    say(f"capacity {cap}")
    nxt = model.prepare(batches[0]) if a.prepare else batches[0]
    for i in range(a.steps):
        cur = nxt
        nxt = model.prepare(batches[(i + 1) % 3]) if a.prepare else batches[(i + 1) % 3]
        say(f"step {i}: prepared")
        loss = model.loss(cur)
        say(f"step {i}: forward done")
        (loss / world).backward()
        say(f"step {i}: backward done")
        model.finish_step()
        say(f"step {i}: update done, loss {float(loss):.6f}")
    ops.raise_if_index_error()
    dist.barrier()
    say("finished")
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--rows", type=int, default=2_000_000, help="table rows per rank")
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--sync", action="store_true")
    ap.add_argument("--prepare", action="store_true")
    ap.add_argument("--stage-host", action="store_true", help="all-to-all buffers cross gloo as host tensors")
    ap.add_argument("--trace", action="store_true", help="synchronise and print around every all-to-all")
    ap.add_argument("--watchdog", type=float, default=60.0)
    a = ap.parse_args()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(a.world, port, a), nprocs=a.world, join=True)
