"""One eager rx_neumf_step call at the MovieLens-1M bench shape (B 8192, 1 + 5 items per sample, D 64), timed
back to back with HIP events.  With a library built under -DNM_PHASE_TIMES (profiles/build_variant.sh) the
kernel prints the phase times of one tile of two workgroups."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from recsys_amd import ops

torch.manual_seed(0)
dev = torch.device("cuda:0")
U, I, D, B, N = 6040, 3706, 64, 8192, 6
p_drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
tabs = [torch.randn(n, D, device=dev) * 0.01 for n in (U, I, U, I)]
w = [torch.randn(64, 128, device=dev) * 0.1, torch.zeros(64, device=dev), torch.randn(32, 64, device=dev) * 0.1, torch.zeros(32, device=dev),
     torch.randn(32, 64, device=dev) * 0.1, torch.zeros(32, device=dev), torch.randn(1, 64, device=dev) * 0.1, torch.zeros(1, device=dev)]
users = torch.randint(0, U, (B,), device=dev)
items = torch.randint(0, I, (B, N), device=dev)
labels = (torch.rand(B, N, device=dev) < 0.2).double()
seed = torch.zeros((), dtype=torch.int64, device=dev)
for _ in range(3):
    out = ops.neumf_step(tabs, users, items, labels, w, p_drop, seed)
torch.cuda.synchronize()
if os.environ.get("NM_ONCE"):
    sys.exit(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 200
e0.record()
for _ in range(K):
    ops.neumf_step(tabs, users, items, labels, w, p_drop, seed)
e1.record()
torch.cuda.synchronize()
print(f"p_drop {p_drop}: rx_neumf_step (both launches) {e0.elapsed_time(e1) / K * 1e3:.1f} us per call, loss {float(out[0]):.6f}")
