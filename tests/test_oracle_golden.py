"""The CPU oracle against the golden vectors captured from the imported reference."""
import numpy as np
import pytest
import torch

from conftest import MODEL_FIXTURES, adagrad_well_conditioned, assert_close, group, load_golden, model_kwargs, to_torch
from oracle.models import POINTWISE, OracleModel


def _build(rec, name):
    kw = model_kwargs(rec)
    return OracleModel(name, to_torch(group(rec, "state")), n_users=kw["n_users"],
                       n_items=kw["n_items"], n_occupations=kw["n_occupations"], dropout=0.0)


@pytest.mark.parametrize("fixture", MODEL_FIXTURES)
def test_forward_loss_grads(fixture):
    rec = load_golden(fixture)
    name = fixture.split("_d")[0]
    batch = to_torch(group(rec, "batch"))
    m = _build(rec, name).eval()
    with torch.no_grad():
        assert_close(m(batch).numpy(), rec["eval_logits"], what="eval logits")
        if name in POINTWISE:
            pw = dict(batch, item_code=batch["item_code"][:, 0].contiguous())
            assert_close(m(pw, pointwise=True).numpy(), rec["pw_logits"], what="pointwise")
    m = _build(rec, name).train()
    loss = m.loss(batch)
    assert loss.dtype == torch.float64
    assert_close(loss.item(), rec["loss"], rtol=1e-6, what="loss")
    loss.backward()
    for k, g in group(rec, "grad").items():
        assert_close(m.params[k].grad.numpy(), g, what=f"grad {k}", atol=1e-7)
    total = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    assert_close(total.item(), rec["clip_total_norm"], what="clip norm")


@pytest.mark.parametrize("fixture", MODEL_FIXTURES)
@pytest.mark.parametrize("opt", ["sgd", "adagrad", "adam"])
def test_optimizer_steps(fixture, opt):
    rec = load_golden(fixture)
    name = fixture.split("_d")[0]
    batch = to_torch(group(rec, "batch"))
    m = _build(rec, name).train()
    make = {"sgd": lambda p: torch.optim.SGD(p, lr=0.1),
            "adagrad": lambda p: torch.optim.Adagrad(p, lr=0.1),
            "adam": lambda p: torch.optim.Adam(p, lr=1e-3)}[opt]
    o = make(m.parameters())
    grads = group(rec, "grad")
    gmax = max(float(np.abs(g).max()) for g in grads.values())
    for s in (1, 2, 3):
        o.zero_grad()
        m.loss(batch).backward()
        if opt == "adam":
            torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        o.step()
        if s in (1, 3):
            sd = m.state_dict()
            for k, v in group(rec, f"{opt}{s}").items():
                if k.endswith("num_batches_tracked"):
                    assert int(sd[k]) == int(v)
                elif k.endswith("running_mean") and opt != "sgd":
                    # tracks the bias of the Linear in front of BatchNorm, whose gradient is
                    # zero in exact arithmetic: under Adagrad / Adam it random-walks on noise
                    continue
                else:
                    a = sd[k].numpy()
                    if opt in ("adagrad", "adam") and k in grads:
                        a, v = adagrad_well_conditioned(a, v, grads[k], gmax)
                    rtol = 1e-5 if s == 1 else (2e-5 if opt == "sgd" else 1e-4)
                    assert_close(a, v, rtol=rtol, what=f"{opt}{s} {k}")


def test_oracle_bce_gradient_matches_torch_op_including_zero_logits():
    """The restated loss must differentiate like aten's binary_cross_entropy_with_logits, also at
    logits that are exactly zero (every ReLU in front of a zero-bias output layer off)."""
    import torch
    from oracle.models import bce_with_logits
    g = torch.Generator().manual_seed(0)
    x = torch.randn(7, 5, generator=g)
    x[0, 0] = x[3, 2] = x[6, 4] = 0.0
    y = (torch.rand(7, 5, generator=g) < 0.4).double()
    a = x.clone().requires_grad_(True)
    b = x.clone().requires_grad_(True)
    la = bce_with_logits(a, y)
    lb = torch.nn.functional.binary_cross_entropy_with_logits(b, y)
    assert la.dtype == lb.dtype == torch.float64 and torch.allclose(la, lb, rtol=1e-7, atol=0)
    la.backward(); lb.backward()
    assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=0)
    assert float(a.grad[0, 0]) == float(b.grad[0, 0])
