import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import bench
from recsys_amd import ops
ops.set_index_check("deferred")
dev = torch.device("cuda", 0)
w = bench.build_workload("fm_criteo", dev, 0, 1, 1.0, True)
for i in range(16): bench.one_step(w, i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200): bench.one_step(w, i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"cpu enqueue {1e6*(t1-t0)/200:.1f} us/step, total {1e6*(t2-t0)/200:.1f} us/step")
g = w["graphed"]
print("nodes?", len(getattr(g, "_graphs", [])))
