"""Host-side profile of the EAGER sharded FM step (what --gpus N > 1 runs by default): cProfile over 50 steps
at one rank, exchanges through RCCL when RX_BENCH_FORCE_DIST=1.  Usage: python profiles/host_profile_sharded.py"""
import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.getcwd())
import bench
from recsys_amd import ops
ops.set_index_check("deferred")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
if os.environ.get("RX_BENCH_FORCE_DIST") == "1":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
w = bench.build_workload("fm_criteo_sharded", dev, 0, 1, 1.0, use_graph=False)
for i in range(10): bench.one_step(w, i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50): bench.one_step(w, i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host issue time {1e3 * (t1 - t0) / 50:.3f} ms per step, with the device drained {1e3 * (t2 - t0) / 50:.3f} ms")
pr = cProfile.Profile(); pr.enable()
for i in range(50): bench.one_step(w, i)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(32)
