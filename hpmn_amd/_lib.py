"""ctypes binding of ``libhpmn_hip.so`` (C ABI declared in ``include/hpmn_hip.h``).

The library is built in-tree by ``hpmn_amd.build.build_library()`` (hipcc, gfx950).  There
is NO fallback: if the shared object is missing or a call fails the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HPMN_MAX_LAYERS = 12
HPMN_ABI_VERSION = 14
HPMN_MAX_RANKS = 8
HPMN_MAX_CHUNKS = 32
HPMN_FWD_NO_CANDIDATE = 1
HPMN_BWD_CANDIDATE_FROM_HS = 1

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libhpmn_hip.so")

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)


class HpmnInputProj(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("T", C.c_int32), ("D", C.c_int32), ("H", C.c_int32),
        ("x", C.c_void_p), ("ids", C.c_void_p), ("emb", C.c_void_p),
        ("Tids", C.c_int32), ("F", C.c_int32), ("E", C.c_int32), ("front_zero", C.c_int32),
        ("mask_id0", C.c_int32),
        ("V", C.c_int64),
        ("wg", C.c_void_p), ("bg", C.c_void_p), ("wc", C.c_void_p), ("bc", C.c_void_p),
        ("xp", C.c_void_p), ("x_out", C.c_void_p),
        ("t_begin", C.c_int32), ("t_len", C.c_int32),
    ]


class HpmnGruFwd(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("T", C.c_int32), ("D", C.c_int32), ("H", C.c_int32),
        ("xp", C.c_void_p), ("wg", C.c_void_p), ("wc", C.c_void_p),
        ("h_last", C.c_void_p), ("h_last_stride", C.c_int64),
        ("y", C.c_void_p), ("period", C.c_int32),
        ("hs", C.c_void_p), ("gates", C.c_void_p),
        ("t_begin", C.c_int32), ("t_end", C.c_int32),
        ("h_init", C.c_void_p), ("h_init_stride", C.c_int64),
    ]


class HpmnGruBwd(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("T", C.c_int32), ("D", C.c_int32), ("H", C.c_int32),
        ("wg", C.c_void_p), ("wc", C.c_void_p),
        ("hs", C.c_void_p), ("gates", C.c_void_p),
        ("d_h_last", C.c_void_p), ("d_h_last_stride", C.c_int64),
        ("d_y", C.c_void_p), ("period", C.c_int32),
        ("d_act", C.c_void_p),
        ("t_begin", C.c_int32), ("t_end", C.c_int32),
        ("dh_carry", C.c_void_p),
        ("d_x", C.c_void_p),
        ("scatter_ids", C.c_void_p), ("d_emb", C.c_void_p), ("d_last", C.c_void_p),
        ("Tids", C.c_int32), ("F", C.c_int32), ("E", C.c_int32), ("front_zero", C.c_int32), ("mask_id0", C.c_int32),
        ("last_t", C.c_int32),
        ("flags", C.c_int32), ("pad_", C.c_int32),
    ]


class HpmnGruWgrad(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("T", C.c_int32), ("D", C.c_int32), ("H", C.c_int32),
        ("x", C.c_void_p), ("hs", C.c_void_p), ("gates", C.c_void_p), ("d_act", C.c_void_p),
        ("wg", C.c_void_p), ("wc", C.c_void_p),
        ("d_wg", C.c_void_p), ("d_bg", C.c_void_p), ("d_wc", C.c_void_p), ("d_bc", C.c_void_p),
        ("d_x", C.c_void_p),
        ("workspace", C.c_void_p),
        ("seq_per_wg", C.c_int32),
        ("t_begin", C.c_int32), ("t_len", C.c_int32),
        ("whole_cu", C.c_int32),
    ]


class HpmnReadDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("K", C.c_int32), ("H", C.c_int32), ("D0", C.c_int32), ("hop", C.c_int32),
        ("off_wq", C.c_int32), ("off_bq", C.c_int32), ("off_map", C.c_int32),
        ("off_att", (C.c_int32 * 6) * 4),
        ("off_gamma", C.c_int32), ("off_beta", C.c_int32),
        ("off_fc", C.c_int32 * 6),
        ("n_params", C.c_int32),
        ("dropout_seed", C.c_uint64),
    ]


class HpmnScanDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("T", C.c_int32), ("F", C.c_int32), ("E", C.c_int32),
        ("H", C.c_int32), ("K", C.c_int32),
        ("front_zero", C.c_int32), ("mask_id0", C.c_int32), ("last_index", C.c_int32),
        ("V", C.c_int64),
        ("periods", C.c_int32 * HPMN_MAX_LAYERS),
    ]


class HpmnTrainStep(C.Structure):
    """One training step behind ONE call (hpmn_train_step, ABI v14)."""
    _fields_ = [
        ("scan", HpmnScanDesc), ("read", HpmnReadDesc),
        ("ids", C.c_void_p), ("label", C.c_void_p),
        ("param", C.c_void_p), ("grad", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p),
        ("n_emb", C.c_int64), ("n_total", C.c_int64),
        ("off_gru", (C.c_int64 * 4) * HPMN_MAX_LAYERS),
        ("off_read", C.c_int64),
        ("memory", C.c_void_p), ("last", C.c_void_p), ("pred", C.c_void_p),
        ("d_memory", C.c_void_p), ("d_last", C.c_void_p),
        ("scan_workspace", C.c_void_p), ("read_workspace", C.c_void_p),
        ("loss_acc", C.c_void_p), ("loss3", C.c_void_p),
        ("mask1", C.c_void_p), ("mask2", C.c_void_p),
        ("keep_prob", C.c_float), ("inv_global_batch", C.c_float), ("memory_reg", C.c_float),
        ("lr_t", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("clip", C.c_float),
        ("clear_grad_first", C.c_int32),
    ]


class HpmnGruFusedFwd(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("T", C.c_int32), ("D", C.c_int32), ("H", C.c_int32),
        ("x", C.c_void_p), ("ids", C.c_void_p), ("emb", C.c_void_p),
        ("Tids", C.c_int32), ("F", C.c_int32), ("E", C.c_int32), ("front_zero", C.c_int32), ("mask_id0", C.c_int32),
        ("V", C.c_int64),
        ("wg", C.c_void_p), ("bg", C.c_void_p), ("wc", C.c_void_p), ("bc", C.c_void_p),
        ("x_out", C.c_void_p),
        ("h_last", C.c_void_p), ("h_last_stride", C.c_int64),
        ("y", C.c_void_p), ("period", C.c_int32),
        ("hs", C.c_void_p), ("gates", C.c_void_p),
        ("last", C.c_void_p), ("last_t", C.c_int32), ("flags", C.c_int32),
    ]


class HpmnGruPairFwd(C.Structure):
    _fields_ = [("lo", HpmnGruFusedFwd), ("up", HpmnGruFusedFwd), ("scratch", C.c_void_p),
                ("img_lo", C.c_void_p), ("img_up", C.c_void_p), ("flags", C.c_int32), ("pad_", C.c_int32)]


class HpmnGruPairBwd(C.Structure):
    _fields_ = [("lo", HpmnGruBwd), ("up", HpmnGruBwd), ("flags", C.c_int32), ("pad_", C.c_int32)]


class HpmnPipe(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("K", C.c_int32), ("H", C.c_int32), ("train", C.c_int32),
        ("T", C.c_int32 * HPMN_MAX_LAYERS), ("D", C.c_int32 * HPMN_MAX_LAYERS), ("period", C.c_int32 * HPMN_MAX_LAYERS),
        ("wg", C.c_void_p * HPMN_MAX_LAYERS), ("bg", C.c_void_p * HPMN_MAX_LAYERS),
        ("wc", C.c_void_p * HPMN_MAX_LAYERS), ("bc", C.c_void_p * HPMN_MAX_LAYERS),
        ("x0", C.c_void_p),
        ("y", C.c_void_p * HPMN_MAX_LAYERS), ("hs", C.c_void_p * HPMN_MAX_LAYERS), ("gates", C.c_void_p * HPMN_MAX_LAYERS),
        ("memory", C.c_void_p), ("d_memory", C.c_void_p),
        ("d_act", C.c_void_p * HPMN_MAX_LAYERS), ("d_x", C.c_void_p * HPMN_MAX_LAYERS),
        ("sync", C.c_void_p),
        ("mem_stride", C.c_int64),
    ]


class HpmnTrainLayout(C.Structure):
    _fields_ = [
        ("K", C.c_int32), ("pad", C.c_int32),
        ("T", C.c_int32 * HPMN_MAX_LAYERS),
        ("x0", C.c_uint64),
        ("xp", C.c_uint64 * HPMN_MAX_LAYERS), ("hs", C.c_uint64 * HPMN_MAX_LAYERS),
        ("gates", C.c_uint64 * HPMN_MAX_LAYERS), ("y", C.c_uint64 * HPMN_MAX_LAYERS),
        ("d_act", C.c_uint64 * HPMN_MAX_LAYERS), ("d_x", C.c_uint64 * HPMN_MAX_LAYERS),
        ("wgrad_ws", C.c_uint64), ("total_bytes", C.c_uint64),
        ("pair_ws", C.c_uint64), ("wgrad_ws_layer", C.c_uint64 * HPMN_MAX_LAYERS),
    ]


class HpmnOnlineUpdate(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("K", C.c_int32),
        ("periods", C.c_int32 * HPMN_MAX_LAYERS),
        ("user", C.c_void_p), ("x", C.c_void_p), ("state", C.c_void_p), ("count", C.c_void_p),
        ("wg", C.c_void_p * HPMN_MAX_LAYERS), ("bg", C.c_void_p * HPMN_MAX_LAYERS),
        ("wc", C.c_void_p * HPMN_MAX_LAYERS), ("bc", C.c_void_p * HPMN_MAX_LAYERS),
    ]


# every symbol include/hpmn_hip.h declares: (restype, argtypes)
class HpmnScatterPlan(C.Structure):
    _fields_ = [("n", C.c_int64), ("perm", C.c_void_p), ("seg", C.c_void_p), ("start", C.c_void_p), ("rows", C.c_void_p),
                ("count", C.c_void_p), ("out_rows", C.c_void_p), ("partials", C.c_void_p)]


class HpmnRowsAdam(C.Structure):
    _fields_ = [("world", C.c_int32), ("E", C.c_int32), ("id_flags", C.c_int32), ("counts_stride", C.c_int32),
                ("ids", C.c_void_p), ("ids_stride", C.c_int64), ("counts", C.c_void_p),
                ("len", C.c_int64 * HPMN_MAX_RANKS), ("first", C.c_int64 * HPMN_MAX_RANKS), ("n", C.c_int64 * HPMN_MAX_RANKS),
                ("rows", C.c_void_p), ("rows_stride", C.c_int64), ("flags", C.c_void_p),
                ("param", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("V", C.c_int64),
                ("lr_t", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("clip", C.c_float),
                ("grad_scale", C.c_float), ("bucket_start", C.c_void_p), ("bucket_stride", C.c_int64),
                ("bucket_shift", C.c_int32), ("pad_", C.c_int32)]


class HpmnTileFwd(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("period", C.c_int32),
                ("pad_", C.c_int32), ("x", C.c_void_p), ("xp", C.c_void_p), ("wg", C.c_void_p), ("bg", C.c_void_p), ("wc", C.c_void_p),
                ("bc", C.c_void_p), ("y", C.c_void_p), ("h_last", C.c_void_p), ("h_last_stride", C.c_int64)]


SIGNATURES = {
    "hpmn_abi_version": (C.c_int, []),
    "hpmn_strerror": (C.c_char_p, [C.c_int]),
    "hpmn_last_hip_error": (C.c_int, []),
    "hpmn_gru_shape_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "hpmn_embed_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                    C.c_int64, C.c_int32, C.c_void_p]),
    "hpmn_gru_input_proj": (C.c_int, [C.POINTER(HpmnInputProj), C.c_void_p]),
    "hpmn_gru_scan_fwd": (C.c_int, [C.POINTER(HpmnGruFwd), C.c_void_p]),
    "hpmn_gru_scan_bwd": (C.c_int, [C.POINTER(HpmnGruBwd), C.c_void_p]),
    "hpmn_gru_scan_bwd_fuses_dx": (C.c_int, [C.c_int32, C.c_int32]),
    "hpmn_gru_candidate_elision": (C.c_int, [C.c_int32, C.c_int32]),
    "hpmn_gru_scan_bwd_fuses_scatter": (C.c_int, [C.c_int32] * 5),
    "hpmn_gru_param_grads_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "hpmn_gru_param_grads": (C.c_int, [C.POINTER(HpmnGruWgrad), C.c_void_p]),
    "hpmn_gru_input_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hpmn_scan_workspace_bytes": (C.c_size_t, [C.POINTER(HpmnScanDesc)]),
    "hpmn_scan_fwd": (C.c_int, [C.POINTER(HpmnScanDesc), C.c_void_p, C.c_void_p,
                                C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hpmn_read_workspace_bytes": (C.c_size_t, [C.POINTER(HpmnReadDesc)]),
    "hpmn_read_workspace_bytes_n": (C.c_size_t, [C.c_int32, C.POINTER(C.POINTER(HpmnReadDesc))]),
    "hpmn_read_fwd": (C.c_int, [C.POINTER(HpmnReadDesc)] + [C.c_void_p] * 8),
    "hpmn_read_fwd_bwd": (C.c_int, [C.POINTER(HpmnReadDesc)] + [C.c_void_p] * 6 +
                          [C.c_float, C.c_float, C.c_float] + [C.c_void_p] * 7),
    "hpmn_read_fwd_n": (C.c_int, [C.c_int32, C.POINTER(C.POINTER(HpmnReadDesc)), C.c_void_p, C.POINTER(C.c_void_p),
                                  C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    "hpmn_read_fwd_bwd_n": (C.c_int, [C.c_int32, C.POINTER(C.POINTER(HpmnReadDesc)), C.c_void_p, C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                      C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "hpmn_read_param_grads": (C.c_int, [C.POINTER(HpmnReadDesc), C.c_void_p, C.c_void_p, C.c_void_p]),
    "hpmn_read_param_grads_n": (C.c_int, [C.c_int32, C.POINTER(C.POINTER(HpmnReadDesc)), C.c_void_p, C.c_void_p, C.c_void_p]),
    "hpmn_read_param_grads_loss_n": (C.c_int, [C.c_int32, C.POINTER(C.POINTER(HpmnReadDesc)), C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "hpmn_embed_grad_scatter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]),
    "hpmn_gru_fused_fwd_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "hpmn_gru_fused_fwd_writes_last": (C.c_int, []),
    "hpmn_gru_fused_fwd": (C.c_int, [C.POINTER(HpmnGruFusedFwd), C.c_void_p]),
    "hpmn_gru_pair_fwd_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "hpmn_gru_pair_fwd_scratch_bytes": (C.c_size_t, []),
    "hpmn_gru_pair_fwd": (C.c_int, [C.POINTER(HpmnGruPairFwd), C.c_void_p]),
    "hpmn_gru_pair_bwd_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "hpmn_gru_pair_bwd": (C.c_int, [C.POINTER(HpmnGruPairBwd), C.c_void_p]),
    "hpmn_gru_proj_image_floats": (C.c_size_t, [C.c_int32]),
    "hpmn_gru_proj_images": (C.c_int, [C.c_int32] + [C.POINTER(C.c_void_p)] * 4 + [C.POINTER(C.c_int32), C.POINTER(C.c_void_p),
                                                                               C.c_void_p]),
    "hpmn_memory_update": (C.c_int, [C.POINTER(HpmnOnlineUpdate), C.c_void_p]),
    "hpmn_adam_step_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                      C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "hpmn_table_mark_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "hpmn_adam_step_table": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                       C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_void_p]),
    "hpmn_scatter_plan_build_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int64]),
    "hpmn_scatter_plan_build": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32,
                                          C.c_void_p, C.c_void_p]),
    "hpmn_tile_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "hpmn_tile_fwd": (C.c_int, [C.POINTER(HpmnTileFwd), C.c_void_p]),
    "hpmn_rows_sum_adam": (C.c_int, [C.POINTER(HpmnRowsAdam), C.c_void_p]),
    "hpmn_table_mark_ranks": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p,
                                        C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "hpmn_train_ctx_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "hpmn_train_ctx_destroy": (None, [C.c_void_p]),
    "hpmn_scan_train_workspace_bytes": (C.c_size_t, [C.POINTER(HpmnScanDesc)]),
    "hpmn_scan_train_layout": (C.c_int, [C.POINTER(HpmnScanDesc), C.POINTER(HpmnTrainLayout)]),
    "hpmn_scan_fwd_train": (C.c_int, [C.c_void_p, C.POINTER(HpmnScanDesc), C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hpmn_scan_bwd": (C.c_int, [C.c_void_p, C.POINTER(HpmnScanDesc), C.c_void_p, C.POINTER(C.c_void_p),
                                C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                                C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p,
                                C.c_void_p, C.c_int32, C.c_void_p]),
    "hpmn_train_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hpmn_train_step": (C.c_int, [C.c_void_p, C.POINTER(HpmnTrainStep), C.c_void_p]),
    "hpmn_train_probe": (C.c_int, [C.c_void_p, C.c_int32]),
    "hpmn_train_probe_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "hpmn_train_mark_layer0_reverse": (C.c_int, [C.c_void_p, C.c_int32]),
    "hpmn_train_wait_layer0_reverse": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hpmn_pipe_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "hpmn_pipe_sync_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "hpmn_pipe_fwd": (C.c_int, [C.POINTER(HpmnPipe), C.c_void_p]),
    "hpmn_pipe_bwd": (C.c_int, [C.POINTER(HpmnPipe), C.c_void_p]),
    "hpmn_embed_gather_seq": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]),
    "hpmn_embed_gather_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                        C.c_int32, C.c_void_p]),
    "hpmn_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_void_p]),
    "hpmn_adam_step_clear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_void_p]),
}


SIGNATURES.update({
    "hpmn_scatter_plan": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hpmn_embed_grad_segsum_partials_floats": (C.c_size_t, [C.c_int64, C.c_int32]),
    "hpmn_embed_grad_segsum_chunk": (C.c_int, []),
    "hpmn_has_legacy_kernels": (C.c_int, []),
    "hpmn_embed_grad_segsum": (C.c_int, [C.POINTER(HpmnScatterPlan), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "hpmn_train_set_scatter_plan": (C.c_int, [C.c_void_p, C.POINTER(HpmnScatterPlan)]),
})


class HpmnLibraryError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load(path: Optional[str] = None) -> C.CDLL:
    """dlopen the HIP library and bind every declared symbol.  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("HPMN_LIB_PATH") or LIB_PATH      # (HPMN_LIB_PATH: developer builds, tools/)
    if not os.path.exists(p):
        raise HpmnLibraryError(
            "%s not found: the HIP extension is not built.  Run `python -c \"import __graft_entry__ as g; "
            "g.build()\"` (or hpmn_amd.build.build_library()).  There is no CPU fallback." % p)
    # PyTorch-ROCm bundles its own libamdhip64; streams and device pointers handed to this library come from
    # THAT runtime.  Import torch first so the library's libamdhip64 dependency resolves to the copy torch has
    # already loaded -- loaded the other way round the process ends up with two HIP runtimes and every launch
    # on a torch stream fails with hipErrorNoDevice (100).
    import torch  # noqa: F401
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.hpmn_abi_version()
    if got != HPMN_ABI_VERSION:
        raise HpmnLibraryError("ABI version mismatch: library %d, binding %d" % (got, HPMN_ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        lib = load()
        msg = lib.hpmn_strerror(rc).decode()
        raise HpmnLibraryError("%s failed: %s (code %d, hipError %d)" % (what, msg, rc, lib.hpmn_last_hip_error()))
