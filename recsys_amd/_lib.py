"""ctypes binding of librecsys_hip.so (the C ABI declared in include/recsys_hip.h).

The library is built in-tree by ``recsys_amd/csrc/Makefile`` (or ``__graft_entry__.build()``)
into ``recsys_amd/librecsys_hip.so``.  There is no fallback: if it is missing, importing this
module raises, and every op in ``recsys_amd`` is unusable.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RX_LIB_PATH: another build of the same library (A/B timing of kernel variants); never a different implementation
LIB_PATH = os.environ.get("RX_LIB_PATH") or os.path.join(_HERE, "librecsys_hip.so")

i32, i64, f32 = C.c_int32, C.c_int64, C.c_float
vp, sz = C.c_void_p, C.c_size_t


class GatherSeg(C.Structure):
    _fields_ = [("table", vp), ("idx", vp), ("rows", i64), ("base", i64), ("dim", i32),
                ("col", i32), ("div", i32), ("stride", i32), ("k", i32), ("bag", i32)]


class MFArgs(C.Structure):
    _fields_ = [("user_table", vp), ("n_users", i64), ("item_table", vp), ("n_items", i64),
                ("occ_table", vp), ("n_occ", i64), ("utemp_table", vp), ("temp_table", vp),
                ("n_temp", i64), ("user_bias", vp), ("item_bias", vp), ("bias", vp),
                ("lin_w", vp), ("lin_b", vp), ("users", vp), ("items", vp), ("occs", vp),
                ("temps", vp), ("B", i64), ("N", i32), ("dim", i32),
                ("loss_parts", vp), ("n_loss_parts", i32), ("pad_", i32), ("loss_count", i64), ("loss_out", vp)]


class FMArgs(C.Structure):
    _fields_ = [("feat", vp), ("bias", vp), ("n_feat", i64), ("ex_idx", vp), ("ex_base", vp),
                ("item_idx", vp), ("item_base", i64), ("ex_ld", i64), ("item_ld", i64),
                ("deep_ld", i64), ("B", i64), ("F", i32), ("N", i32),
                ("dim", i32), ("flags", i32), ("feat_ld", i64), ("bias_ld", i64)]


RX_FG_MAX_FIELDS = 64


class FGArgs(C.Structure):
    _fields_ = [("ex_idx", vp), ("item_idx", vp), ("ex_ld", i64), ("item_ld", i64), ("B", i64),
                ("n_fields", i32), ("pad_", i32), ("field_base", i64 * (RX_FG_MAX_FIELDS + 1))]


class FGReduceArgs(C.Structure):
    _fields_ = [("T", vp), ("ld_T", i64), ("g", vp), ("ld_g", i64), ("deep", vp), ("deep_ld", i64), ("B", i64),
                ("n_fields", i32), ("dim", i32), ("mode", i32), ("premultiplied", i32), ("srow", vp), ("sb", vp),
                ("target", vp), ("state", vp), ("table", vp), ("target2", vp), ("state2", vp), ("n_rows", i64),
                ("lr", f32), ("eps", f32), ("scale", vp), ("sq_part", vp), ("ld_w", i64), ("ld_w2", i64),
                ("undo", vp), ("undo2", vp),
                ("loss_parts", vp), ("n_loss_parts", i32), ("pad2_", i32), ("loss_count", i64), ("loss_out", vp)]




class NeuMFArgs(C.Structure):
    _fields_ = [("user_gmf", vp), ("item_gmf", vp), ("user_mlp", vp), ("item_mlp", vp), ("n_users", i64), ("n_items", i64),
                ("users", vp), ("items", vp), ("labels", vp), ("labels_f64", i32), ("N", i32), ("B", i64), ("dim", i32),
                ("pad_", i32), ("w1", vp), ("b1", vp), ("w2", vp), ("b2", vp), ("wg", vp), ("bg", vp), ("wf", vp), ("bf", vp),
                ("p_drop", f32), ("pad2_", f32), ("seed", vp), ("du_gmf", vp), ("du_mlp", vp), ("di_gmf", vp), ("di_mlp", vp),
                ("wgrad", vp), ("loss", vp), ("logits", vp)]


RX_FM_NO_SECOND_ORDER = 1
RX_MULTI_MAX = 24


class TensorList(C.Structure):
    _fields_ = [("w", vp * RX_MULTI_MAX), ("g", vp * RX_MULTI_MAX), ("state", vp * RX_MULTI_MAX),
                ("n", i64 * RX_MULTI_MAX), ("count", i32)]


class AdamList(C.Structure):
    _fields_ = [("w", vp * RX_MULTI_MAX), ("g", vp * RX_MULTI_MAX), ("m", vp * RX_MULTI_MAX), ("v", vp * RX_MULTI_MAX),
                ("n", i64 * RX_MULTI_MAX), ("count", i32)]


RX_SEG_MULTI_MAX = 8
RX_BATCH_MAX_TENSORS = 8


class BatchPlan(C.Structure):
    _fields_ = [("src", vp * RX_BATCH_MAX_TENSORS), ("dst", vp * RX_BATCH_MAX_TENSORS),
                ("row_bytes", i64 * RX_BATCH_MAX_TENSORS), ("count", i32), ("rows_per_batch", i64), ("n_batches", i64),
                ("ahead", i64)]


class SegTable(C.Structure):
    _fields_ = [("occ", vp), ("occ_ld", i64), ("n", i64), ("dim", i32),
                ("sorted_pos", vp), ("sorted_slot", vp), ("uniq_rows", vp), ("seg_off", vp), ("n_uniq", vp),
                ("target", vp), ("state", vp), ("occ2", vp), ("target2", vp), ("state2", vp),
                ("n_rows", i64), ("sq_part", vp)]


# name -> (restype, argtypes); must list every symbol of include/recsys_hip.h
PROTOTYPES = {
    "rx_abi_version": (C.c_int, []),
    "rx_last_error": (C.c_char_p, []),
    "rx_init": (C.c_int, [C.c_int]),
    "rx_wall_clock": (C.c_int, [vp, C.c_int, vp]),
    "rx_gather_concat": (C.c_int, [C.POINTER(GatherSeg), C.c_int, vp, i64, i64, vp, C.c_int, vp]),
    "rx_gather_concat_bwd": (C.c_int, [vp, i64, i64, i32, i32, i32, vp, C.c_int, vp]),
    "rx_mf_score_fwd": (C.c_int, [C.POINTER(MFArgs), vp, vp, C.c_int, vp]),
    "rx_mf_score_fwd_bce_workspace_bytes": (sz, [i64, i32]),
    "rx_mf_score_fwd_bce_partials": (i32, [i64, i32]),
    "rx_mf_score_fwd_bce": (C.c_int, [C.POINTER(MFArgs), vp, C.c_int, vp, vp, vp, vp, sz, vp, C.c_int, vp]),
    "rx_mf_bwd_grid": (i64, [i64, i32]),
    "rx_mf_score_bwd": (C.c_int, [C.POINTER(MFArgs), vp, vp, vp, vp, vp, vp, vp, C.c_int, vp]),
    "rx_colsum_partials": (C.c_int, [vp, i64, i32, vp, C.c_int, vp]),
    "rx_gmf_mul_fwd": (C.c_int, [vp, i64, vp, i64, vp, vp, i64, i32, i32, vp, i64, i32, vp, C.c_int, vp]),
    "rx_gmf_mul_bwd": (C.c_int, [vp, i64, vp, i64, vp, vp, i64, i32, i32, vp, i64, i32, vp, vp, C.c_int, vp]),
    "rx_fm2_fwd": (C.c_int, [C.POINTER(FMArgs), vp, vp, vp, vp, C.c_int, vp]),
    "rx_fm2_bwd": (C.c_int, [C.POINTER(FMArgs), vp, vp, vp, vp, C.c_int, vp]),
    "rx_fm2_flat_indices": (C.c_int, [C.POINTER(FMArgs), vp, C.c_int, vp]),
    "rx_bce_workspace_bytes": (sz, [i64]),
    "rx_bce_fwd": (C.c_int, [vp, vp, C.c_int, i64, vp, vp, vp, sz, C.c_int, vp]),
    "rx_bce_fwd_noalloc": (C.c_int, [vp, vp, C.c_int, i64, vp, vp, C.c_int, vp]),
    "rx_neumf_step_supported": (C.c_int, [i32, i32]),
    "rx_neumf_step_weight_floats": (i64, []),
    "rx_neumf_step_workspace_bytes": (sz, [i64, i32]),
    "rx_neumf_step": (C.c_int, [C.POINTER(NeuMFArgs), vp, sz, vp, C.c_int, vp]),
    "rx_sort_dedup_workspace_bytes": (sz, [i64]),
    "rx_sort_dedup": (C.c_int, [vp, i64, i64, vp, vp, vp, vp, vp, vp, vp, sz, vp, C.c_int, vp]),
    "rx_sort_dedup_fm": (C.c_int, [C.POINTER(FMArgs), vp, vp, vp, vp, vp, vp, vp, sz, vp, C.c_int, vp]),
    "rx_segreduce_grid": (i64, [i64, i32]),
    "rx_segreduce_workspace_bytes": (sz, [i64, i32]),
    "rx_segreduce": (C.c_int, [vp, i64, i64, i32, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, i64, f32, f32, vp, vp, vp, sz, C.c_int, vp]),
    "rx_segreduce_fm": (C.c_int, [vp, i64, vp, vp, i64, i64, i32, i32, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, i64, f32, f32, vp, vp, vp, sz, C.c_int, vp]),
    "rx_fg_supported": (C.c_int, [C.POINTER(FGArgs)]),
    "rx_fg_dim_supported": (C.c_int, [i32]),
    "rx_fg_workspace_bytes": (sz, [i64, i32]),
    "rx_fg_group": (C.c_int, [C.POINTER(FGArgs), vp, vp, vp, sz, vp, C.c_int, vp]),
    "rx_fg_reduce_grid": (i64, [i64, i32]),
    "rx_fg_reduce_workspace_bytes": (sz, [i64, i32]),
    "rx_fg_reduce": (C.c_int, [C.POINTER(FGReduceArgs), vp, sz, C.c_int, vp]),
    "rx_fm2_fwd_bce_workspace_bytes": (sz, [i64, i32]),
    "rx_fm2_fwd_bce_partials": (i64, [i64, i32]),
    "rx_fm2_fwd_bce": (C.c_int, [C.POINTER(FMArgs), vp, C.c_int, vp, vp, i64, vp, vp, sz, vp, C.c_int, vp]),
    "rx_fg_row_update": (C.c_int, [vp, i32, vp, i64, C.c_int, vp, vp, vp, vp, vp, i64, f32, f32, vp, i64, i64, C.c_int, vp]),
    "rx_fg_row_update_clip": (C.c_int, [vp, i32, vp, i64, C.c_int, vp, vp, vp, vp, vp, i64, f32, f32, vp, i64, f32, vp, vp,
                                        i64, i64, C.c_int, vp]),
    "rx_row_update": (C.c_int, [vp, i32, vp, vp, i64, C.c_int, vp, vp, vp, vp, vp, i64, f32, f32, vp, C.c_int, vp]),
    "rx_epoch_advance": (C.c_int, [C.POINTER(BatchPlan), vp, vp, vp, C.c_int, vp, C.c_int, vp]),
    "rx_segreduce_multi": (C.c_int, [C.POINTER(SegTable), i32, C.c_int, f32, f32, vp, vp, sz, C.c_int, vp]),
    "rx_row_update_multi": (C.c_int, [C.POINTER(SegTable), i32, C.c_int, f32, f32, vp, C.c_int, vp]),
    "rx_dense_adam": (C.c_int, [vp, vp, vp, vp, i64, f32, f32, f32, f32, i32, vp, C.c_int, vp]),
    "rx_bn_apply16": (C.c_int, [vp, i64, i64, i32, vp, vp, vp, vp, C.c_int, vp, f32, vp, i64, C.c_int, vp]),
    "rx_bn_bwd16": (C.c_int, [vp, i64, vp, i64, vp, i64, i64, i32, vp, vp, vp, C.c_int, f32, C.c_int, vp, i64, vp, vp, vp, sz,
                              C.c_int, vp]),
    "rx_to_bf16": (C.c_int, [vp, i64, vp, i64, i64, i32, C.c_int, C.c_int, vp]),
    "rx_linear16_fwd": (C.c_int, [vp, i64, vp, vp, vp, vp, i64, i64, i32, i32, C.c_int, C.c_int, C.c_int, vp]),
    "rx_linear16_bwd_x": (C.c_int, [vp, i64, vp, vp, vp, i64, vp, i64, i64, i32, i32, C.c_int, C.c_int, C.c_int, vp]),
    "rx_linear16_bwd_w_workspace_bytes": (sz, [i64, i32, i32]),
    "rx_linear16_bwd_w": (C.c_int, [vp, i64, vp, i64, vp, vp, i64, i32, i32, vp, sz, C.c_int, vp]),
    "rx_cross16_fwd": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i64, i32, C.c_int, vp]),
    "rx_cross16_layer_fwd": (C.c_int, [vp, i64, vp, vp, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i64, i32, C.c_int, vp]),
    "rx_cross16_bwd": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i64, i32, C.c_int, C.c_int, C.c_int, vp]),
    "rx_sample_negatives": (C.c_int, [vp, vp, vp, i64, i64, i32, i32, C.c_uint64, vp, vp, vp, C.c_int, vp]),
    "rx_dense_adam_dstep": (C.c_int, [vp, vp, vp, vp, i64, f32, f32, f32, f32, vp, vp, C.c_int, vp]),
    "rx_dense_sgd": (C.c_int, [vp, vp, i64, f32, vp, C.c_int, vp]),
    "rx_dense_adagrad": (C.c_int, [vp, vp, vp, i64, f32, f32, vp, C.c_int, vp]),
    "rx_sumsq_workspace_bytes": (sz, [i64]),
    "rx_sumsq_accumulate": (C.c_int, [vp, i64, vp, vp, sz, C.c_int, vp]),
    "rx_sum_accumulate": (C.c_int, [vp, i64, vp, vp, sz, C.c_int, vp]),
    "rx_clip_scale": (C.c_int, [vp, f32, vp, vp, C.c_int, vp]),
    "rx_bn_workspace_bytes": (C.c_size_t, [i64, i32]),
    "rx_bn_stats": (C.c_int, [vp, i64, i64, i32, f32, f32, vp, vp, vp, vp, vp, C.c_size_t, C.c_int, vp]),
    "rx_bn_apply": (C.c_int, [vp, i64, i64, i32, vp, vp, vp, vp, C.c_int, vp, f32, vp, i64, C.c_int, vp]),
    "rx_bn_bwd": (C.c_int, [vp, i64, vp, i64, vp, i64, i64, i32, vp, vp, vp, C.c_int, f32, C.c_int, vp, i64, vp, vp,
                            vp, C.c_size_t, C.c_int, vp]),
    "rx_clip_scale_parts": (C.c_int, [vp, i64, vp, i64, vp, f32, vp, vp, C.c_int, vp]),
    "rx_multi_sumsq_partials": (C.c_size_t, [i32]),
    "rx_multi_sumsq": (C.c_int, [C.POINTER(TensorList), vp, C.c_int, vp]),
    "rx_multi_dense_update": (C.c_int, [C.POINTER(TensorList), C.c_int, f32, f32, vp, C.c_int, vp]),
    "rx_step_tail": (C.c_int, [C.POINTER(SegTable), i32, C.c_int, f32, f32, C.POINTER(TensorList), C.c_int, f32, f32, vp, i64, vp, i64, f32,
                               vp, vp, C.c_int, vp]),
    "rx_multi_dense_adam": (C.c_int, [C.POINTER(AdamList), f32, f32, f32, f32, i32, vp, vp, C.c_int, vp]),
    "rx_reduce_sum": (C.c_int, [vp, i64, vp, vp, sz, C.c_int, vp]),
    "rx_linear_fwd": (C.c_int, [vp, i64, vp, vp, vp, i64, i64, i32, i32, C.c_int, C.c_int, C.c_int, vp]),
    "rx_linear_bwd_x": (C.c_int, [vp, i64, vp, vp, i64, i64, i32, i32, C.c_int, C.c_int, C.c_int, vp]),
    "rx_linear_bwd_w_workspace_bytes": (sz, [i64, i32, i32]),
    "rx_linear_bwd_w": (C.c_int, [vp, i64, vp, i64, vp, vp, i64, i32, i32, C.c_int, vp, sz, C.c_int, vp]),
    "rx_relu_bwd": (C.c_int, [vp, vp, vp, i64, C.c_int, vp]),
    "rx_relu_bwd_scaled": (C.c_int, [vp, vp, vp, i64, f32, C.c_int, vp]),
    "rx_cross_combine_fwd": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, i64, i64, i32, C.c_int, vp]),
    "rx_cross_combine_bwd": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i64, i32, C.c_int, vp]),
    "rx_cross_combine_bwd_stack": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i64, i32, C.c_int, C.c_int,
                                             C.c_int, vp]),
    "rx_bucketize_workspace_bytes": (sz, [i64, i32]),
    "rx_bucketize_by_owner": (C.c_int, [vp, i64, vp, i32, i64, vp, vp, vp, vp, sz, vp, C.c_int, vp]),
    "rx_unpermute_rows": (C.c_int, [vp, vp, i64, vp, i32, vp, C.c_int, vp]),
    "rx_permute_rows": (C.c_int, [vp, vp, i64, vp, i32, vp, C.c_int, vp]),
    "rx_pack_by_owner": (C.c_int, [vp, vp, vp, i32, i64, i64, i64, vp, vp, vp, C.c_int, vp]),
    "rx_zero_unused_slots": (C.c_int, [vp, i32, i64, i32, vp, vp, C.c_int, vp]),
    "rx_fm_flat_rows": (C.c_int, [vp, vp, i64, i32, vp, C.c_int, vp]),
    "rx_compose_slots": (C.c_int, [vp, vp, i64, i64, vp, vp, C.c_int, vp]),
    "rx_topk_codes": (C.c_int, [vp, vp, i64, i32, i32, vp, C.c_int, vp]),
}

RX_SEG_DENSE, RX_SEG_COMPACT, RX_SEG_SGD, RX_SEG_ADAGRAD, RX_SEG_ROWWISE_ADAGRAD, RX_SEG_SGD_SPEC, RX_SEG_SGD_SPEC_REDO = range(7)
RX_SEG_SGD_SPEC_OLD, RX_SEG_COMPACT_IF_CLIPPED, RX_SEG_SGD_IF_CLIPPED, RX_SEG_RESTORE, RX_SEG_RESTORE_IF_CLIPPED = range(7, 12)
RX_ERR_INDEX = 1
RX_ERR_CAPACITY = 2
RX_ABI_VERSION = 1


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). recsys_amd has no "
            "fallback path; its ops run only through this HIP library."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    got = lib.rx_abi_version()
    if got != RX_ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {got}, expected {RX_ABI_VERSION}; rebuild it")
    return lib


lib = _load()


_initialised = set()


def init_device(device: int) -> None:
    """Per-device set-up of the library (rx_init): resolves the addresses of its own device arrays.
    Must run outside stream capture; ``ops`` calls it the first time a device is used."""
    if device not in _initialised:
        check(lib.rx_init(int(device)), "rx_init")
        _initialised.add(device)


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib.rx_last_error()
        raise RuntimeError(f"{what} failed: {msg.decode() if msg else 'unknown error'}")
