#!/usr/bin/env python3
"""Randomised cross-checks of the index and reduce kernels against torch ops on the same device.
Not part of the test suite (run time is a parameter): python profiles/fuzz_gpu.py [seconds] [seed].
Every case draws a shape, an index distribution and a mode; failures print the case and exit 1."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
gen = torch.Generator().manual_seed(seed)


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=gen))


def draw_indices(n, rows):
    kind = ri(0, 3)
    if kind == 0:
        idx = torch.randint(0, rows, (n,), generator=gen)
    elif kind == 1:      # heavy head
        idx = (torch.rand(n, generator=gen) ** 6 * rows).long().clamp_(0, rows - 1)
    elif kind == 2:      # a handful of rows only
        idx = torch.randint(0, min(rows, ri(1, 5)), (n,), generator=gen)
    else:                # last rows of the table
        idx = rows - 1 - torch.randint(0, min(rows, 50), (n,), generator=gen)
    return idx


def close(a, b, what, rtol=1e-5):
    scale = float(b.abs().max()) if b.numel() else 1.0
    err = float((a.double() - b.double()).abs().max()) if b.numel() else 0.0
    if not err <= rtol * max(scale, 1e-30) + 1e-12:
        raise AssertionError(f"{what}: max err {err:.3e} at scale {scale:.3e}")


def case_sort():
    n = ri(1, 3) ** 4 * ri(1, 40000) // 8 + 1
    rows = [ri(1, 50), ri(50, 100000), ri(10**5, 4 * 10**7), 2**32][ri(0, 3)]
    idx = draw_indices(n, min(rows, 2**40))
    si = ops.sort_dedup(idx.to(dev), rows, want_inverse=True)
    nu = int(si.n_uniq.item())
    exp_pos = torch.sort(idx, stable=True).indices
    uniq, inv, counts = torch.unique(idx, sorted=True, return_inverse=True, return_counts=True)
    assert nu == uniq.numel(), "n_uniq"
    assert torch.equal(si.sorted_pos.cpu().long(), exp_pos), "sorted_pos"
    assert torch.equal(si.uniq_rows[:nu].cpu(), uniq), "uniq_rows"
    seg = si.seg_off[:nu + 1].cpu().long()
    assert torch.equal(seg[1:] - seg[:-1], counts), "seg_off"
    assert torch.equal(si.inverse.cpu().long(), inv), "inverse"
    assert torch.equal(si.sorted_slot.cpu().long(), inv[exp_pos]), "sorted_slot"
    return f"sort n={n} rows={rows}"


def case_segreduce():
    n, rows = ri(1, 60000), ri(1, 30000)
    dim = [1, 3, 4, 8, 16, 20, 32, 64, 128, 256][ri(0, 9)]
    idx = draw_indices(n, rows).to(dev)
    occ = torch.randn(n, dim, generator=gen).to(dev)
    si = ops.sort_dedup(idx, rows)
    mode = [_lib.RX_SEG_DENSE, _lib.RX_SEG_SGD, _lib.RX_SEG_ADAGRAD][ri(0, 2)]
    with2 = ri(0, 1) == 1
    occ2 = torch.randn(n, generator=gen).to(dev) if with2 else None
    G = torch.zeros(rows, dim, dtype=torch.float64, device=dev).index_add_(0, idx, occ.double())
    G2 = torch.zeros(rows, dtype=torch.float64, device=dev).index_add_(0, idx, occ2.double()) if with2 else None
    w0 = torch.randn(rows, dim, generator=gen).to(dev)
    b0 = torch.randn(rows, 1, generator=gen).to(dev)
    if mode == _lib.RX_SEG_DENSE:
        t, t2 = torch.zeros(rows, dim, device=dev), (torch.zeros(rows, 1, device=dev) if with2 else None)
        ops.segreduce(occ, si, mode, t, occ2=occ2, target2=t2)
        close(t, G, "dense grad")
        if with2:
            close(t2.reshape(-1), G2, "dense companion grad")
    elif mode == _lib.RX_SEG_SGD:
        w, b = w0.clone(), b0.clone()
        ops.segreduce(occ, si, mode, w, lr=0.1, occ2=occ2, target2=b if with2 else None)
        close(w, w0.double() - 0.1 * G, "sgd rows")
        if with2:
            close(b.reshape(-1), b0.reshape(-1).double() - 0.1 * G2, "sgd companion")
    else:
        w, st = w0.clone(), torch.zeros(rows, dim, device=dev)
        b, st2 = b0.clone(), torch.zeros(rows, 1, device=dev)
        ops.segreduce(occ, si, mode, w, state=st, lr=0.1, eps=1e-10, occ2=occ2, target2=b if with2 else None,
                      state2=st2 if with2 else None)
        touched = G.abs().sum(1) > 0
        ref = w0.double() - 0.1 * G / ((G * G).sqrt() + 1e-10)
        ref[~touched] = w0[~touched].double()
        close(w[touched], ref[touched], "adagrad rows", rtol=2e-5)
        close(st, G * G, "adagrad state", rtol=2e-5)
    return f"segreduce n={n} rows={rows} dim={dim} mode={mode} companion={with2}"


def case_fm():
    B, F = ri(1, 3000), ri(1, 30)
    dim = [4, 8, 16, 32, 64, 128][ri(0, 5)]
    dims = [ri(1, 3000) for _ in range(F + 1)]
    base = torch.tensor([sum(dims[:f]) for f in range(F)]).to(dev)
    item_base, n_feat = sum(dims[:F]), sum(dims)
    ex = torch.stack([draw_indices(B, d) for d in dims[:F]], 1).to(dev)
    item = draw_indices(B, dims[F]).reshape(B, 1).to(dev)
    feat = (torch.randn(n_feat, dim, generator=gen) * 0.3).to(dev)
    bias = torch.randn(n_feat, generator=gen).to(dev)
    logits, _, tsum = ops.fm2_fwd(feat, bias, ex, base, item, item_base, want_tsum=True)
    rows = torch.cat([ex + base[None, :], item + item_base], 1)
    e = feat[rows].double()
    s = e.sum(1)
    ref = bias[rows].double().sum(1, keepdim=True) + 0.5 * (s * s - (e * e).sum(1)).sum(-1, keepdim=True)
    close(logits, ref, "fm logits")
    close(tsum, s, "fm row sums")
    a = ops.fm_sort_dedup(feat, ex, base, item, item_base)
    flat = ops.fm2_flat_indices(feat, ex, base, item, item_base)
    b = ops.sort_dedup(flat, n_feat)
    nu = int(a.n_uniq.item())
    assert nu == int(b.n_uniq.item()) and torch.equal(a.sorted_pos, b.sorted_pos) and torch.equal(a.uniq_rows[:nu], b.uniq_rows[:nu])
    g = torch.randn(B, generator=gen).to(dev) / B
    grad = torch.zeros(n_feat, dim, device=dev)
    grad2 = torch.zeros(n_feat, 1, device=dev)
    ops.segreduce_fm(tsum, g, F, a, _lib.RX_SEG_DENSE, grad, table=feat, target2=grad2)
    per_occ = g.double()[:, None, None] * (s[:, None, :] - e)                    # [B, F+1, D]
    order = torch.cat([rows[:, :F].reshape(-1), rows[:, F]])
    vals = torch.cat([per_occ[:, :F].reshape(-1, dim), per_occ[:, F]])
    G = torch.zeros(n_feat, dim, dtype=torch.float64, device=dev).index_add_(0, order, vals)
    close(grad, G, "fm fused gradient", rtol=2e-5)
    return f"fm B={B} F={F} dim={dim}"


def case_owner():
    n, world = ri(1, 50000), ri(1, 8)
    rows = ri(world, 10**6)
    idx = draw_indices(n, rows)
    counts, send_local, perm = ops.bucketize_by_owner(idx.to(dev), world, rows)
    owner = idx % world
    exp_perm = torch.sort(owner, stable=True).indices
    assert torch.equal(perm.cpu().long(), exp_perm), "owner permutation"
    assert torch.equal(counts.cpu().long(), torch.bincount(owner, minlength=world)), "owner counts"
    assert torch.equal(send_local.cpu(), (idx // world)[exp_perm]), "local row ids"
    return f"owner n={n} world={world}"


def case_linear():
    M = [ri(1, 300), ri(300, 5000), 128 * ri(1, 40)][ri(0, 2)]
    N = [1, ri(1, 70), 4 * ri(1, 130), 128 * ri(1, 4)][ri(0, 3)]
    K = [ri(1, 70), 4 * ri(1, 130), 16 * ri(1, 30)][ri(0, 2)]
    bf16 = ri(0, 1) == 1
    x = torch.randn(M, K, generator=gen).to(dev)
    w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=gen).to(dev)
    dy = torch.randn(M, N, generator=gen).to(dev)
    act = ri(0, 1)
    if bf16:   # products of bf16-rounded operands are exact in fp32: compare on rounded operands
        xr, wr, dyr = (t.bfloat16().float() for t in (x, w, dy))
    else:
        xr, wr, dyr = x, w, dy
    y = ops.linear_fwd(x, w, b, act, bf16)
    ref = xr.double() @ wr.double().T + b.double()
    close(y, ref.clamp_min(0) if act else ref, f"linear fwd bf16={bf16}", rtol=2e-5)
    dx = ops.linear_bwd_x(dy, w, bf16)
    close(dx[:, :K], dyr.double() @ wr.double(), f"linear grad-x bf16={bf16}", rtol=2e-5)
    dw, db = ops.linear_bwd_w(dy, x, True, bf16)
    close(dw, dyr.double().T @ xr.double(), f"linear grad-w bf16={bf16}", rtol=2e-5)
    close(db, dy.double().sum(0), "linear grad-b", rtol=2e-5)
    return f"linear M={M} N={N} K={K} bf16={bf16} act={act}"


def case_topk():
    rows, n, k = ri(1, 400), ri(1, 100), 0
    k = ri(1, min(n, 20))
    scores = torch.randn(rows, n, generator=gen)
    scores[torch.rand(rows, n, generator=gen) < 0.2] = 0.5          # ties: the larger column comes first
    codes = torch.randint(0, 5000, (rows, n), generator=gen)
    got = ops.topk_codes(scores.to(dev), codes.to(dev), k).cpu()
    order = torch.sort(scores, dim=1, stable=True).indices.flip(1)[:, :k]   # stable ascending argsort read backwards
    assert torch.equal(got, torch.gather(codes, 1, order)), "top-k codes"
    return f"topk rows={rows} n={n} k={k}"


CASES = [case_sort, case_segreduce, case_fm, case_owner, case_linear, case_topk]
ops.set_index_check("sync")
t0, done = time.time(), 0
while time.time() - t0 < budget:
    fn = CASES[ri(0, len(CASES) - 1)]
    state = gen.get_state()
    try:
        desc = fn()
        ops.raise_if_index_error()
    except Exception as exc:   # report the case, then fail
        print(f"FAILED after {done} cases in {fn.__name__}: {exc!r}  (generator state saved to gpurun_out/fuzz_state.pt)")
        os.makedirs("gpurun_out", exist_ok=True)
        torch.save(state, "gpurun_out/fuzz_state.pt")
        raise
    done += 1
    if done % 50 == 0:
        print(f"{done} cases ok, last: {desc}", flush=True)
print(f"{done} randomised cases passed in {time.time() - t0:.0f} s (seed {seed})")
