#!/usr/bin/env python3
"""Back-to-back timing of the field-grouped backward's ops on the fm_criteo batch shape (B = 65536,
26 fields, 33 M rows): rx_fg_group, rx_fg_reduce (COMPACT + norm partials), rx_fg_row_update and
rx_fm2_fwd, each re-issued REPS times between two HIP events.  Run under
`rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recsys_amd import _lib, ops, synthetic  # noqa: E402

REPS = int(os.environ.get("REPS", 20))
D = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
dims = synthetic.criteo_field_dims() if D == 16 else synthetic.criteo_field_dims(total_rows=125_000_000)
B, nf = 65536, len(dims)
base = [0]
for d in dims:
    base.append(base[-1] + d)
torch.manual_seed(0)
table = torch.randn(base[-1], D, device=dev)
bias = torch.randn(base[-1], device=dev)
batches = [synthetic.criteo_batch(dims, B, seed=i, device=dev)["sparse"] for i in range(4)]
ex_base = torch.tensor(base[:nf - 1], device=dev)
T = torch.randn(B, D, device=dev)
g = torch.randn(B, device=dev) / B
ops.set_index_check("deferred")


def timed(name, fn):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(REPS):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    print(f"{name:16s} {1e3 * s.elapsed_time(e) / REPS:8.1f} us")


fgs = [ops.fg_group(sp[:, :nf - 1], sp[:, nf - 1:], base) for sp in batches]
timed("fg_group", lambda i: ops.fg_group(batches[i % 4][:, :nf - 1], batches[i % 4][:, nf - 1:], base, out=fgs[i % 4]))
comp = torch.empty(B * nf, D, device=dev)
comp2 = torch.empty(B * nf, device=dev)
sq = torch.empty(ops.fg_reduce_grid(B * nf, D), device=dev)
timed("fg_reduce", lambda i: ops.fg_reduce(T, g, fgs[i % 4], _lib.RX_SEG_COMPACT, comp, table=table, target2=comp2, want_sq=True, sq_out=sq))
scale = torch.zeros(1, device=dev)
timed("fg_row_update", lambda i: ops.fg_row_update(comp, fgs[i % 4], _lib.RX_SEG_SGD, table, lr=0.05, scale=scale, compact2=comp2, table2=bias))
timed("fg_row_update/nobias", lambda i: ops.fg_row_update(comp, fgs[i % 4], _lib.RX_SEG_SGD, table, lr=0.05, scale=scale))
timed("fg_reduce/nobias", lambda i: ops.fg_reduce(T, g, fgs[i % 4], _lib.RX_SEG_COMPACT, comp, table=table, want_sq=True, sq_out=sq))
timed("fm2_fwd/nobias", lambda i: ops.fm2_fwd(table, None, batches[i % 4][:, :nf - 1], ex_base, batches[i % 4][:, nf - 1:], base[nf - 1], want_tsum=True))
timed("fm2_fwd", lambda i: ops.fm2_fwd(table, bias, batches[i % 4][:, :nf - 1], ex_base, batches[i % 4][:, nf - 1:], base[nf - 1], want_tsum=True))
labels = (torch.rand(B, 1, device=dev) < 0.25).double()
timed("fm2_fwd_bce", lambda i: ops.fm2_fwd_bce(table, bias, batches[i % 4][:, :nf - 1], ex_base, batches[i % 4][:, nf - 1:], base[nf - 1], labels))
timed("bce", lambda i: ops.bce_with_logits(T[:, :1].contiguous(), labels))
if D == 16:
    # the same three ops on rows that sit in 128-byte lines of their own, bias entry behind the row
    buf = torch.zeros(base[-1], 32, device=dev)
    buf[:, :16].copy_(table)
    buf[:, 16].copy_(bias)
    ltab, lbias = buf[:, :16], buf[:, 16]
    timed("line/fm2_fwd_bce", lambda i: ops.fm2_fwd_bce(ltab, lbias, batches[i % 4][:, :nf - 1], ex_base, batches[i % 4][:, nf - 1:], base[nf - 1], labels))
    timed("line/fg_reduce", lambda i: ops.fg_reduce(T, g, fgs[i % 4], _lib.RX_SEG_COMPACT, comp, table=ltab, target2=comp2, want_sq=True, sq_out=sq))
    timed("line/fg_row_update", lambda i: ops.fg_row_update(comp, fgs[i % 4], _lib.RX_SEG_SGD, ltab, lr=0.05, scale=scale, compact2=comp2, table2=lbias))
    # the clipped SGD step taken inside the reduce (undo records beside), and the second launch when the clip does
    # not bite (every block forms the factor, finds 1 and ends) / when it does (rows redone from the records)
    undo = torch.empty(B * nf, 32, device=dev)
    timed("line/fg_reduce_spec", lambda i: ops.fg_reduce(T, g, fgs[i % 4], _lib.RX_SEG_SGD_SPEC, ltab, target2=lbias, lr=1e-9, want_sq=True, sq_out=sq, undo=undo))
    timed("line/fg_update_spec_idle", lambda i: ops.fg_row_update(undo, fgs[i % 4], _lib.RX_SEG_SGD_SPEC, ltab, lr=1e-9, table2=lbias, clip_parts=sq, max_norm=1e30))
    timed("line/fg_update_spec_redo", lambda i: ops.fg_row_update(undo, fgs[i % 4], _lib.RX_SEG_SGD_SPEC, ltab, lr=1e-9, table2=lbias, clip_parts=sq, max_norm=1e-30))
    if os.environ.get("ADJ") == "1":     # experiment: 256-byte row stride, the undo record in the line behind the row's (needs a -DFG_EXP_ADJ build)
        del buf, ltab, lbias
        buf4 = torch.zeros(base[-1], 64, device=dev)
        buf4[:, :16].copy_(table)
        atab, abias = buf4[:, :16], buf4[:, 16]
        timed("adj/fm2_fwd_bce", lambda i: ops.fm2_fwd_bce(atab, abias, batches[i % 4][:, :nf - 1], ex_base, batches[i % 4][:, nf - 1:], base[nf - 1], labels))
        timed("adj/fg_reduce", lambda i: ops.fg_reduce(T, g, fgs[i % 4], _lib.RX_SEG_COMPACT, comp, table=atab, target2=comp2, want_sq=True, sq_out=sq))
        timed("adj/fg_reduce_spec", lambda i: ops.fg_reduce(T, g, fgs[i % 4], _lib.RX_SEG_SGD_SPEC, atab, target2=abias, lr=1e-9, want_sq=True, sq_out=sq, undo=undo))
if os.environ.get("SORTED") == "1":
    # the same ops on occurrence lists ordered by ascending row inside every field (any order that keeps a
    # row's occurrences together is valid input): what the bin order costs in memory locality
    import dataclasses
    sfgs = []
    for sp, fg in zip(batches, fgs):
        rows, sbs = [], []
        for f in range(nf):
            r, order = torch.sort(sp[:, f], stable=True)
            rows.append(r + base[f]); sbs.append(order)
        srow = torch.cat(rows).to(torch.int32)
        sfgs.append(dataclasses.replace(fg, srow=srow, sb=torch.cat(sbs).to(torch.int32)))
    timed("sorted/fg_reduce", lambda i: ops.fg_reduce(T, g, sfgs[i % 4], _lib.RX_SEG_COMPACT, comp, table=table, target2=comp2, want_sq=True, sq_out=sq))
    timed("sorted/fg_row_update", lambda i: ops.fg_row_update(comp, sfgs[i % 4], _lib.RX_SEG_SGD, table, lr=0.05, scale=scale, compact2=comp2, table2=bias))
    timed("sorted/fg_row_update/nobias", lambda i: ops.fg_row_update(comp, sfgs[i % 4], _lib.RX_SEG_SGD, table, lr=0.05, scale=scale))
if os.environ.get("OLD") == "1":
    sis = [ops.fm_sort_dedup(table, sp[:, :nf - 1], ex_base, sp[:, nf - 1:], base[nf - 1]) for sp in batches]
    timed("fm_sort_dedup", lambda i: ops.fm_sort_dedup(table, batches[i % 4][:, :nf - 1], ex_base, batches[i % 4][:, nf - 1:], base[nf - 1], out=sis[i % 4]))
ops.raise_if_index_error()
