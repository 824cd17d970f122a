import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


def load_golden(name):
    """Load a fixture as {key: ndarray}; npz only, pickles refused."""
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def group(rec, prefix):
    p = prefix + "/"
    return {k[len(p):]: v for k, v in rec.items() if k.startswith(p)}


def to_torch(d, device="cpu"):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in d.items()}


MODEL_FIXTURES = [
    "VanillaMF_d16", "VanillaMF_d64", "BiasMF_d16", "SideFeaturesMF_d16", "TemporalMF_d16",
    "FactorizationMachine_d16", "FactorizationMachine_d64", "GMF_d16", "MLP_d16", "NeuMF_d16",
    "NeuMF_d64", "WidenDeep_d16", "DeepFactorizationMachine_d16",
]


def model_kwargs(rec):
    n_users, n_items, n_occ, max_ts, dim, b, n = (int(x) for x in rec["dims"])
    return dict(n_users=n_users, n_items=n_items, n_occupations=n_occ,
                max_timestamp_rank=max_ts, embedding_dim=dim)


ELEMENTWISE_FLOOR = 1e-3     # entries at least this fraction of max|expected| are also held element-wise
ELEMENTWISE_FACTOR = 100.0   # ... to ELEMENTWISE_FACTOR * rtol relative to the entry itself
WORST = {}                   # category -> largest element-wise error / rtol seen this session


def _category(what: str, rtol: float) -> str:
    """Which class of comparison a check belongs to (for the end-of-run report): the 3-step optimizer
    states are compared at 1e-4 and would otherwise hide how close logits and gradients come."""
    w = what.lower()
    if rtol >= 1e-3:
        return "bf16 operands"
    for tag in ("sgd3", "adagrad3", "adam3"):
        if tag in w:
            return "weights after 3 steps"
    for tag in ("sgd1", "adagrad1", "adam1", "updated", "rows", "after", "weight", "tower ", "deep_mlp."):
        if tag in w and "grad " not in w:
            return "weights after 1-2 steps"
    if "grad" in w or "norm" in w:
        return "gradients"
    if "logit" in w:
        return "logits"
    if "loss" in w:
        return "loss"
    return "other (kernels against torch / numpy)"


def assert_close(actual, expected, rtol=1e-5, what="", atol=0.0, elem_rtol=None):
    """Norm-wise bound: |a - e| <= rtol * max(|e|, scale) for every element, scale = max|expected|
    of the tensor -- accuracy relative to the tensor's magnitude, so that sums with cancellation
    are not held to a relative bound on a near-zero entry.  On top of that, entries with
    |e| >= ELEMENTWISE_FLOOR * scale are held element-wise to |a - e| <= elem_rtol * |e|
    (default ELEMENTWISE_FACTOR * rtol; ``elem_rtol=False`` switches it off, for comparisons across
    number formats such as bf16 operands against an fp32 reference, where an entry that cancels to
    1e-3 of the scale carries no correct digit by construction), and the worst such ratio of the
    session is reported at the end of the run.  ``atol`` is for quantities that are zero in exact arithmetic (e.g. the gradient
    of a Linear bias that feeds BatchNorm), where both sides hold rounding noise only."""
    a = np.asarray(actual, dtype=np.float64)
    e = np.asarray(expected, dtype=np.float64)
    assert a.shape == e.shape, f"{what}: shape {a.shape} vs {e.shape}"
    scale = float(np.abs(e).max()) if e.size else 0.0
    d = np.abs(a - e)
    tol = rtol * np.maximum(np.abs(e), scale) + atol + 1e-12
    bad = d > tol
    if bad.any():
        i = np.argmax(d - tol)
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{a.size} off; worst |d|={d.flat[i]:.3e} "
            f"at {i} (a={a.flat[i]:.8e}, e={e.flat[i]:.8e}, scale={scale:.3e}, rtol={rtol})"
        )
    big = np.abs(e) >= ELEMENTWISE_FLOOR * scale
    if elem_rtol is not False and scale > 0 and big.any():
        rel = np.where(big, np.maximum(d - atol, 0.0) / np.where(big, np.abs(e), 1.0), 0.0)
        i = int(np.argmax(rel))
        worst = float(rel.flat[i])
        cat = _category(what, rtol)
        if rtol > 0 and worst / rtol > WORST.get(cat, {"ratio": 0.0})["ratio"]:
            WORST[cat] = dict(ratio=worst / rtol, what=what, rel=worst, rtol=rtol)
        limit = ELEMENTWISE_FACTOR * rtol if elem_rtol is None else elem_rtol
        if worst > limit:
            raise AssertionError(
                f"{what}: element-wise relative error {worst:.3e} > {limit:.1e} at {i} "
                f"(a={a.flat[i]:.8e}, e={e.flat[i]:.8e}, |e|/scale={abs(e.flat[i]) / scale:.2e})")


def pytest_terminal_summary(terminalreporter):
    if WORST:
        terminalreporter.write_line(
            f"assert_close: worst element-wise relative error on entries >= {ELEMENTWISE_FLOOR:g} * max|e|, per class of comparison:")
        for cat, w in sorted(WORST.items(), key=lambda kv: -kv[1]["ratio"]):
            terminalreporter.write_line(
                f"  {cat:38s} {w['rel']:.3e} = {w['ratio']:6.1f} x rtol ({w['rtol']:g}) in '{w['what']}'")


def adagrad_well_conditioned(actual, expected, grad, gmax=None, floor=1e-4):
    """Adagrad's and Adam's first steps move a weight by about lr * g / |g|: where g is
    rounding noise around zero the step is decided by that noise (on either side of the
    comparison), so elements with |g| below ``floor`` times the largest gradient entry of the
    model are left out.  Rows of a 2-D table whose gradient row is exactly zero (rows the
    batch never touched) are kept: they must come back unchanged.  Returns the kept elements."""
    g = np.abs(np.asarray(grad, dtype=np.float64))
    keep = g > floor * (g.max() if gmax is None else gmax)
    if g.ndim == 2:
        keep |= (g.max(axis=1, keepdims=True) == 0)
    return np.asarray(actual)[keep], np.asarray(expected)[keep]
