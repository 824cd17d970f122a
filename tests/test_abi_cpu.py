"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol the header
declares, and the host wrappers reject bad tensors before any launch (no GPU needed)."""
import os
import re

import pytest
import torch

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "recsys_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from recsys_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(_lib.lib, s), f"{s} declared in recsys_hip.h but not exported"
        assert s in _lib.PROTOTYPES, f"{s} has no ctypes prototype"
    assert set(_lib.PROTOTYPES) == set(syms)
    assert _lib.lib.rx_abi_version() == _lib.RX_ABI_VERSION


def test_struct_layouts_match_header(tmp_path):
    """The header is plain C; sizeof/offsetof from gcc must equal the ctypes mirrors."""
    import ctypes as C
    import subprocess
    from recsys_amd import _lib
    src = tmp_path / "abi.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "recsys_hip.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(rx_gather_seg), sizeof(rx_mf_args), sizeof(rx_fm_args),'
        ' offsetof(rx_gather_seg, dim), offsetof(rx_mf_args, B), offsetof(rx_fm_args, flags),'
        ' sizeof(rx_tensor_list), offsetof(rx_tensor_list, count), offsetof(rx_gather_seg, bag),'
        ' sizeof(rx_fg_args), offsetof(rx_fg_args, field_base), sizeof(rx_fg_reduce_args), offsetof(rx_fg_reduce_args, srow),'
        ' offsetof(rx_fg_reduce_args, sq_part), sizeof(rx_seg_table), offsetof(rx_seg_table, sorted_pos), offsetof(rx_seg_table, sq_part), sizeof(rx_batch_plan), offsetof(rx_batch_plan, count), offsetof(rx_batch_plan, n_batches));return 0;}\n')
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    exp = [C.sizeof(_lib.GatherSeg), C.sizeof(_lib.MFArgs), C.sizeof(_lib.FMArgs), _lib.GatherSeg.dim.offset,
           _lib.MFArgs.B.offset, _lib.FMArgs.flags.offset, C.sizeof(_lib.TensorList), _lib.TensorList.count.offset,
           _lib.GatherSeg.bag.offset, C.sizeof(_lib.FGArgs), _lib.FGArgs.field_base.offset, C.sizeof(_lib.FGReduceArgs),
           _lib.FGReduceArgs.srow.offset, _lib.FGReduceArgs.sq_part.offset, C.sizeof(_lib.SegTable),
           _lib.SegTable.sorted_pos.offset, _lib.SegTable.sq_part.offset, C.sizeof(_lib.BatchPlan),
           _lib.BatchPlan.count.offset, _lib.BatchPlan.n_batches.offset]
    assert got == exp


def test_workspace_queries_are_pure_host_calls():
    from recsys_amd import _lib
    assert _lib.lib.rx_sort_dedup_workspace_bytes(0) > 0
    a, b = _lib.lib.rx_sort_dedup_workspace_bytes(1000), _lib.lib.rx_sort_dedup_workspace_bytes(100000)
    assert 16 * 1000 <= a < b
    assert _lib.lib.rx_bce_workspace_bytes(10) >= 8
    assert _lib.lib.rx_segreduce_grid(1024, 16) == 1 + 1   # 32 pieces: 4 lanes each in pass 1, 16 pieces per wavefront in pass 2
    assert _lib.lib.rx_segreduce_grid(1024, 64) == 2 + 2
    assert _lib.lib.rx_segreduce_workspace_bytes(1024, 16) == 32 * 2 * 16 * 4 + 256 + 256
    assert _lib.lib.rx_mf_bwd_grid(1024, 64) == 64


def test_wrappers_reject_cpu_and_wrong_dtype_tensors():
    from recsys_amd import ops
    U, V = torch.randn(4, 8), torch.randn(5, 8)
    users, items = torch.zeros(3, dtype=torch.int64), torch.zeros(3, 2, dtype=torch.int64)
    with pytest.raises(ValueError, match="GPU"):
        ops.mf_score_fwd(U, V, users, items)
    with pytest.raises(ValueError, match="GPU"):
        ops.sort_dedup(users, 4)
    with pytest.raises(TypeError):
        ops.mf_score_fwd("not a tensor", V, users, items)


def test_models_keep_reference_parameter_names():
    from conftest import MODEL_FIXTURES, group, load_golden, model_kwargs
    import inspect
    from recsys_amd.models import matrix_factorization as mf, neuralcf as ncf
    for fx in MODEL_FIXTURES:
        rec = load_golden(fx)
        name = fx.split("_d")[0]
        cls = getattr(mf, name, None) or getattr(ncf, name)
        kw = {k: v for k, v in model_kwargs(rec).items() if k in inspect.signature(cls.__init__).parameters}
        m = cls(**kw)
        ref = group(rec, "state")
        sd = m.state_dict()
        assert list(sd.keys()) == list(ref.keys()), name
        for k, v in ref.items():
            assert tuple(sd[k].shape) == tuple(v.shape), (name, k)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in ref.items()})
        sparse = list(m.sparse_parameters())
        assert sparse and all(p.requires_grad for p in sparse)


def test_same_seed_gives_reference_initial_weights():
    """Construction order and init calls mirror the reference, so the golden initial state (made
    with torch.manual_seed(100 + i)) is reproduced bit for bit."""
    from conftest import MODEL_FIXTURES, group, load_golden, model_kwargs
    import inspect
    from recsys_amd.models import matrix_factorization as mf, neuralcf as ncf
    order = ["VanillaMF", "BiasMF", "SideFeaturesMF", "TemporalMF", "FactorizationMachine", "GMF", "MLP",
             "NeuMF", "WidenDeep", "DeepFactorizationMachine"]
    for fx in MODEL_FIXTURES:
        rec = load_golden(fx)
        name = fx.split("_d")[0]
        cls = getattr(mf, name, None) or getattr(ncf, name)
        kw = {k: v for k, v in model_kwargs(rec).items() if k in inspect.signature(cls.__init__).parameters}
        torch.manual_seed(100 + order.index(name))
        m = cls(**kw)
        for k, v in group(rec, "state").items():
            assert torch.equal(m.state_dict()[k], torch.from_numpy(v)), (name, k)
