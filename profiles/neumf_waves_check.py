"""Compares the NeuMF one-launch step run with four and with eight wavefronts per workgroup (RX_NEUMF_WAVES),
and explains rows that differ: a first-layer unit whose pre-activation is zero to rounding passes its gradient
in one summation order and not in the other.  Usage: python profiles/neumf_waves_check.py [run W out.pt]"""
import os
import subprocess
import sys

import torch


def run(out):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from recsys_amd import synthetic
    from recsys_amd.models.neuralcf import NeuMF
    d = synthetic.ML1M
    B, N = 8192, 5
    g = torch.Generator().manual_seed(B * 100 + N)
    batch = {"user_code": torch.randint(0, d["n_users"], (B,), generator=g).cuda(),
             "item_code": torch.randint(0, d["n_items"], (B, N), generator=g).cuda(),
             "label": (torch.rand(B, N, generator=g) < 0.3).double().cuda()}
    torch.manual_seed(3)
    m = NeuMF(d["n_users"], d["n_items"], 64, dropout=0.0).cuda().train()
    with torch.no_grad():
        for e in (m.user_embedding_gmf, m.item_embedding_gmf, m.user_embedding_mlp, m.item_embedding_mlp):
            e.weight.mul_(30.0)
        for lin in (m.linear_mlp[0], m.linear_mlp[3], m.linear_gmf, m.linear_final):
            lin.bias.normal_(0, 0.1)
    loss = m.loss(batch)
    loss.backward()
    torch.save({"loss": loss.item(), "grads": {k: p.grad.cpu() for k, p in m.named_parameters()},
                "params": {k: p.detach().cpu() for k, p in m.named_parameters()},
                "batch": {k: v.cpu() for k, v in batch.items()}}, out)


def main():
    outs = {}
    for w in ("4", "8"):
        out = f"/tmp/nm_w{w}.pt"
        subprocess.run([sys.executable, __file__, "run", out], check=True, env=dict(os.environ, RX_NEUMF_WAVES=w))
        outs[w] = torch.load(out)
    a, b = outs["4"], outs["8"]
    print("loss", a["loss"], b["loss"])
    for k in a["grads"]:
        ga, gb = a["grads"][k], b["grads"][k]
        diff = (ga - gb).abs()
        print(f"{k}: max|d| {diff.max().item():.3e} of max {ga.abs().max().item():.3e}; entries off by > 1e-5 max: "
              f"{int((diff > 1e-5 * ga.abs().max()).sum())}")
    # pre-activations of the first tower layer in fp64: which (pair, unit) sits at zero to fp32 rounding?
    p, bt = a["params"], a["batch"]
    u = bt["user_code"][:, None].expand_as(bt["item_code"]).reshape(-1)
    it = bt["item_code"].reshape(-1)
    x = torch.cat([p["user_embedding_mlp.weight"][u], p["item_embedding_mlp.weight"][it]], 1).double()
    z = x @ p["linear_mlp.0.weight"].double().t() + p["linear_mlp.0.bias"].double()
    near = (z.abs() < 3e-7).nonzero()
    for pr, unit in near.tolist():
        print(f"pair {pr} (user {int(u[pr])}, item {int(it[pr])}) unit {unit}: z = {z[pr, unit].item():.3e}")
    gi = (a["grads"]["user_embedding_mlp.weight"] - b["grads"]["user_embedding_mlp.weight"]).abs().sum(1).nonzero().flatten()
    big = [int(r) for r in gi if (a["grads"]["user_embedding_mlp.weight"][r] - b["grads"]["user_embedding_mlp.weight"][r]).abs().max() > 1e-9]
    print("user rows whose MLP gradient differs by more than 1e-9:", big)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        main()
