"""The multi-field oracle reduces to the reference FM: a 4-field FieldFM fed the reference's
[user, occupation, timestamp | item] indices reproduces the golden FactorizationMachine logits
for the first item of every row."""
import numpy as np
import torch

from conftest import assert_close, group, load_golden, model_kwargs, to_torch
from oracle.fields import field_fm_logits


def test_field_fm_equals_reference_fm_on_four_fields():
    rec = load_golden("FactorizationMachine_d16")
    kw = model_kwargs(rec)
    P = to_torch(group(rec, "state"))
    b = to_torch(group(rec, "batch"))
    # reference table order: users | items | occupations | timestamp ranks.  FieldFM wants the
    # "item" field last, so present the blocks as fields [user, occ, ts, item] of a re-ordered table.
    nu, ni, no, nt = kw["n_users"], kw["n_items"], kw["n_occupations"], kw["max_timestamp_rank"]
    perm = torch.cat([torch.arange(0, nu), torch.arange(nu + ni, nu + ni + no + nt), torch.arange(nu, nu + ni)])
    P2 = {k: v[perm] for k, v in P.items()}
    for n in range(b["item_code"].shape[1]):
        sparse = torch.stack([b["user_code"], b["user_occupation"], b["item_timestamp_rank"], b["item_code"][:, n]], 1)
        out = field_fm_logits(P2, {"sparse": sparse}, [nu, no, nt, ni])
        assert_close(out[:, 0].numpy(), rec["eval_logits"][:, n], what=f"item column {n}")
