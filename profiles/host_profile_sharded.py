import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.getcwd())
import bench
from recsys_amd import ops
ops.set_index_check("deferred")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
os.environ["RX_BENCH_SHARDED_ROWS"] = "33000000"
w = bench.build_workload("fm_criteo_sharded", dev, 0, 1, None, use_graph=False)
for i in range(10): bench.one_step(w, i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(50): bench.one_step(w, i)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
