#!/usr/bin/env python3
"""Randomised model-level cross-checks: a random reference model class at random sizes (batch,
items per row, embedding width, table sizes) on the GPU against the CPU restatement under oracle/:
training loss, every parameter gradient, evaluation logits.  The oracle is the checker only.
Usage: python tests/tools/fuzz_models.py [seconds] [seed]"""
import inspect
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.models import OracleModel, init_state  # noqa: E402
from recsys_amd.models import matrix_factorization as mf, neuralcf as ncf  # noqa: E402

NAMES = ["VanillaMF", "BiasMF", "SideFeaturesMF", "TemporalMF", "FactorizationMachine", "GMF", "MLP", "NeuMF",
         "WidenDeep", "DeepFactorizationMachine"]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
gen = torch.Generator().manual_seed(seed)
ONLY = int(os.environ["FUZZ_ONLY_CASE"]) if "FUZZ_ONLY_CASE" in os.environ else None   # replay one case, with an fp64 referee


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=gen))


def close(a, b, what, rtol=1e-5, atol=1e-7):
    scale = float(b.abs().max()) if b.numel() else 1.0
    d = (a.double() - b.double()).abs()
    err = float(d.max()) if b.numel() else 0.0
    if not err <= rtol * scale + atol:
        where = ""
        if b.dim() == 2:
            rows = d.max(1).values
            top = torch.topk(rows, min(5, rows.numel()))
            where = f"; worst rows {top.indices.tolist()} errs {[f'{v:.2e}' for v in top.values.tolist()]}; " \
                    f"rows above rtol: {int((rows > rtol * scale + atol).sum())}/{rows.numel()}; " \
                    f"|b| of worst row {float(b[top.indices[0]].abs().max()):.3e}"
        raise AssertionError(f"{what}: max err {err:.3e} at scale {scale:.3e}{where}")


t0, done = time.time(), 0
while time.time() - t0 < budget:
    name = NAMES[ri(0, len(NAMES) - 1)]
    dim = [4, 8, 12, 16, 20, 32, 64, 128][ri(0, 7)]
    if name in ("MLP", "NeuMF", "GMF", "WidenDeep", "DeepFactorizationMachine"):
        dim = [8, 16, 32, 64][ri(0, 3)]        # tower widths are derived from dim (dim/2, dim/4 ...)
    B, N = ri(2, 700), [1, 2, 5, 7, 100][ri(0, 4)]
    if N == 100:
        B = ri(2, 40)
    sizes = dict(n_users=ri(3, 3000), n_items=ri(3, 2000), n_occ=ri(1, 30), max_ts=ri(2, 500))
    case = f"{name} dim={dim} B={B} N={N} {sizes} seed={seed} case={done}"
    state_seed = ri(0, 10**6)
    draw = lambda n, hi: (torch.rand(n, generator=gen) ** ri(1, 4) * hi).long().clamp_(0, hi - 1)  # noqa: E731
    batch = {"user_code": draw(B, sizes["n_users"]), "item_code": draw(B * N, sizes["n_items"]).reshape(B, N),
             "user_occupation": draw(B, sizes["n_occ"]), "item_timestamp_rank": draw(B, sizes["max_ts"]),
             "label": (torch.rand(B, N, generator=gen) < 0.3).double()}
    if ONLY is not None and done != ONLY:      # replaying the draws up to one chosen case
        done += 1
        continue
    try:
        state = init_state(name, n_users=sizes["n_users"], n_items=sizes["n_items"], embedding_dim=dim,
                           n_occupations=sizes["n_occ"], max_timestamp_rank=sizes["max_ts"], seed=state_seed)
        oracle = OracleModel(name, state, n_users=sizes["n_users"], n_items=sizes["n_items"],
                             n_occupations=sizes["n_occ"], dropout=0.0)
        cls = getattr(mf, name, None) or getattr(ncf, name)
        kw = dict(n_users=sizes["n_users"], n_items=sizes["n_items"], n_occupations=sizes["n_occ"],
                  max_timestamp_rank=sizes["max_ts"], embedding_dim=dim)
        kw = {k: v for k, v in kw.items() if k in inspect.signature(cls.__init__).parameters}
        m = cls(**kw)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        m.load_state_dict(state)
        m = m.cuda()
        dbatch = {k: v.cuda() for k, v in batch.items()}
        oracle.train(); m.train()
        if ONLY is not None:
            # who is closer to the truth?  the same restatement in fp64 referees the two fp32 results
            ref64 = OracleModel(name, {k: (v.double() if v.is_floating_point() else v) for k, v in state.items()},
                                n_users=sizes["n_users"], n_items=sizes["n_items"], n_occupations=sizes["n_occ"], dropout=0.0)
            ref64.train()
            ref64.loss(batch).backward()
            o32 = OracleModel(name, state, n_users=sizes["n_users"], n_items=sizes["n_items"], n_occupations=sizes["n_occ"], dropout=0.0)
            o32.train(); o32.loss(batch).backward()
            m.train(); m.loss(dbatch).backward()
            gp = dict(m.named_parameters())
            for k, p in ref64.params.items():
                if p.grad is None:
                    continue
                sc = float(p.grad.abs().max()) + 1e-30
                print(f"  {k:40s} scale {sc:.2e}  fp32 CPU restatement err {float((o32.params[k].grad.double() - p.grad).abs().max()) / sc:.2e}"
                      f"  GPU err {float((gp[k].grad.cpu().double() - p.grad).abs().max()) / sc:.2e}")
            print("fp64 referee done for", case)
            break
        lo, lm = oracle.loss(batch), m.loss(dbatch)
        bn_model = name in ("WidenDeep", "DeepFactorizationMachine")
        close(lm.detach().cpu(), lo.detach(), "loss", rtol=1e-5 if bn_model else 2e-6, atol=0)   # batch statistics over few rows amplify rounding
        lo.backward(); lm.backward()
        params = dict(m.named_parameters())
        for k, p in oracle.params.items():
            # BatchNorm towers amplify rounding in front of the normalisation: their Linear biases
            # get gradients that are pure rounding noise
            if p.grad is None or (name in ("WidenDeep", "DeepFactorizationMachine") and k.endswith("bias") and "mlp" in k):
                continue
            # BatchNorm + ReLU towers: an activation that lands within rounding of the ReLU kink (|y| ~ 1e-7,
            # about one case in a hundred at these sizes) may be gated differently by two correct
            # implementations, which moves that column's beta gradient by a fraction of a percent and
            # everything upstream with it.  Their gradients are therefore only checked for gross errors
            # here (FUZZ_ONLY_CASE replays a case with an fp64 referee); the golden tests pin them tightly.
            if bn_model:
                a, b = params[k].grad.cpu().double(), p.grad.double()
                l2 = float((a - b).norm() / (b.norm() + 1e-30))
                if l2 > 0.1:
                    raise AssertionError(f"grad {k}: relative L2 error {l2:.2e}")
            elif name in ("MLP", "NeuMF"):
                # ReLU towers: a pre-activation within rounding of zero may be gated differently by two
                # correct implementations; that moves the gradient rows of the one sample involved
                a, b = params[k].grad.cpu().double(), p.grad.double()
                a2, b2 = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
                bad = ((a2 - b2).abs().max(1).values > 2e-5 * float(b.abs().max()) + 1e-7).sum()
                l2 = float((a - b).norm() / (b.norm() + 1e-30))
                if ("embedding" in k and int(bad) > max(2, a2.shape[0] // 100)) or ("embedding" not in k and l2 > 2e-2):
                    close(params[k].grad.cpu(), p.grad, f"grad {k} ({int(bad)} rows off, relative L2 {l2:.1e})", rtol=2e-5)
            else:
                close(params[k].grad.cpu(), p.grad, f"grad {k}", rtol=2e-5)
        oracle.eval(); m.eval()
        with torch.no_grad():
            close(m(dbatch).cpu(), oracle(batch), "eval logits")
    except Exception as exc:
        print(f"FAILED: {case}: {exc!r}")
        raise
    done += 1
    if done % 20 == 0:
        print(f"{done} cases ok, last: {case}", flush=True)
print(f"{done} randomised model cases passed in {time.time() - t0:.0f} s (seed {seed})")
