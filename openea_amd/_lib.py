"""ctypes binding of libopenea_hip.so (the C ABI declared in include/openea_hip.h).

The product path has NO CPU fallback: if the library is missing, cannot be loaded, or no
HIP device is visible, every compute entry point raises ``OpenEAHipError``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OPENEA_HIP_LIB: load another build of the same library (kernel experiments); default = the in-tree build.
# OEA_STEP_DETERMINISTIC=1: the build whose translational step accumulates its gradients in int64 fixed point
# (libopenea_hip_det.so, csrc/common.h): the same bits run to run and for any number of ranks.
DETERMINISTIC = os.environ.get("OEA_STEP_DETERMINISTIC", "0")[:1] == "1"
LIB_PATH = os.environ.get("OPENEA_HIP_LIB") or os.path.join(_HERE, "csrc",
                                                            "libopenea_hip_det.so" if DETERMINISTIC else "libopenea_hip.so")


class OpenEAHipError(RuntimeError):
    pass


class StepCfg(C.Structure):
    """mirror of `oea_step_cfg` (include/openea_hip.h)."""
    _fields_ = [("loss_kind", C.c_int32), ("l1", C.c_int32), ("margin", C.c_float),
                ("pos_margin", C.c_float), ("neg_margin", C.c_float), ("balance", C.c_float),
                ("ent_l2_norm", C.c_int32), ("rel_l2_norm", C.c_int32), ("opt_kind", C.c_int32),
                ("lr", C.c_float), ("neg_group_k", C.c_int32), ("score_kind", C.c_int32),
                ("normal", C.c_void_p), ("normal_acc", C.c_void_p),
                ("ent_transfer_base", C.c_int32), ("rel_transfer_base", C.c_int32),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("opt_t", C.c_int32)]


class SamplerSide(C.Structure):
    """mirror of `oea_sampler_side` (include/openea_hip.h)."""
    _fields_ = [("table", C.c_void_p), ("capacity", C.c_uint64), ("entity_list", C.c_void_p),
                ("ent_pos", C.c_void_p), ("nbr", C.c_void_p), ("n_ent_list", C.c_int32), ("nbr_k", C.c_int32),
                ("filter", C.c_void_p), ("filter_bits", C.c_uint64)]


class CsrSplit(C.Structure):
    """mirror of `oea_csr_split` (include/openea_hip.h)."""
    _fields_ = [("chunk_row", C.c_void_p), ("chunk_e0", C.c_void_p), ("chunk_e1", C.c_void_p), ("rows", C.c_void_p),
                ("n_chunks", C.c_int32), ("n_rows", C.c_int32), ("threshold", C.c_int32),
                ("row_chunk0", C.c_void_p), ("partials", C.c_void_p), ("partials_floats", C.c_int64), ("tickets", C.c_void_p)]


class AttnGraph(C.Structure):
    """mirror of `oea_attn_graph` (include/openea_hip.h)."""
    _fields_ = [("sub_ptr", C.c_void_p), ("sub_seg", C.c_void_p), ("seg_sub_ptr", C.c_void_p), ("seg_row", C.c_void_p),
                ("colidx", C.c_void_p), ("n_sub", C.c_int64), ("n_seg", C.c_int64), ("nnz", C.c_int64),
                ("agg_rowptr", C.c_void_p), ("agg_colidx", C.c_void_p), ("agg_edge", C.c_void_p), ("agg_rows", C.c_int64),
                ("agg_split", C.POINTER(CsrSplit)),
                ("t_rowptr", C.c_void_p), ("t_row", C.c_void_p), ("t_edge", C.c_void_p), ("t_rows", C.c_int64),
                ("t_split", C.POINTER(CsrSplit)),
                ("sub0", C.c_int64), ("sub1", C.c_int64), ("seg0", C.c_int64), ("seg1", C.c_int64),
                ("agg_row0", C.c_int64), ("agg_row1", C.c_int64), ("agg_slot0", C.c_int64), ("agg_slot1", C.c_int64),
                ("t_row0", C.c_int64), ("t_row1", C.c_int64), ("t_slot0", C.c_int64), ("t_slot1", C.c_int64)]


class GcnUnit(C.Structure):
    """mirror of `oea_gcn_unit` (include/openea_hip.h)."""
    _fields_ = ([(pre + suf, t) for pre in ("a_", "at_", "f_", "ft_")
                 for suf, t in (("rowptr", C.c_void_p), ("colidx", C.c_void_p), ("vals", C.c_void_p), ("split", C.POINTER(CsrSplit)))]
                + [(nm, C.c_void_p) for nm in ("row_ids", "ill", "neg_left", "neg_right", "neg2_left", "neg2_right", "pair_rowptr",
                                               "pair_other", "pair_slot")]
                + [("n", C.c_int64), ("w_rows", C.c_int64), ("t", C.c_int64), ("dim", C.c_int32), ("ld", C.c_int32), ("k", C.c_int32),
                   ("gamma", C.c_float), ("lr", C.c_float)])


class GcnUnitBuffers(C.Structure):
    """mirror of `oea_gcn_unit_buffers`."""
    _fields_ = [(nm, C.c_void_p) for nm in ("t", "x", "h1", "out", "g_out", "g_pre1", "g_x", "g_t", "coef")]


class RotateCfg(C.Structure):
    """mirror of `oea_rotate_cfg` (include/openea_hip.h)."""
    _fields_ = [("gamma", C.c_double), ("phase_scale", C.c_double), ("lr", C.c_double), ("beta1", C.c_double),
                ("beta2", C.c_double), ("eps", C.c_double), ("t", C.c_int64), ("ent_l2_norm", C.c_int32),
                ("rel_l2_norm", C.c_int32), ("opt_kind", C.c_int32), ("reserved", C.c_int32)]


LOSS_KIND = {"margin-based": 0, "limited": 1, "logistic": 2, "positive": 3, "align": 4}
OPT_KIND = {"SGD": 0, "Adagrad": 1, "Adam": 2, "Adadelta": 3}
METRIC = {"inner": 0, "manhattan": 1, "euclidean": 2, "manhattan_f32": 3}

_vp, _i32, _i64, _u32, _u64, _f32, _sz = (C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64,
                                          C.c_float, C.c_size_t)

# name -> (restype, argtypes); every symbol include/openea_hip.h declares
PROTOTYPES = {
    "oea_version": (C.c_int, []),
    "oea_last_error": (C.c_char_p, []),
    "oea_device_count": (C.c_int, []),
    "oea_profile_begin": (C.c_int, [_i32]),
    "oea_profile_end": (C.c_int, [_i32, C.POINTER(C.c_double), C.POINTER(_i32)]),
    "oea_store_create": (C.c_int, [_i64, _i32, C.POINTER(_vp)]),
    "oea_store_destroy": (C.c_int, [_vp]),
    "oea_store_rows": (_i64, [_vp]),
    "oea_store_dim": (_i32, [_vp]),
    "oea_store_ld": (_i32, [_vp]),
    "oea_store_rows_ptr": (_vp, [_vp]),
    "oea_store_load_host": (C.c_int, [_vp, _vp, _vp]),
    "oea_store_save_host": (C.c_int, [_vp, _vp, _vp]),
    "oea_gather_rows": (C.c_int, [_vp, _i32, _i32, _vp, _i64, _i32, _vp, _i32, _vp]),
    "oea_normalize_rows": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp]),
    "oea_fill_f32": (C.c_int, [_vp, _i64, _f32, _vp]),
    "oea_copy_to_host": (C.c_int, [_vp, _vp, _sz, _vp]),
    "oea_copy_from_host": (C.c_int, [_vp, _vp, _sz, _vp]),
    "oea_step_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "oea_triple_step": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64,
                                  C.POINTER(StepCfg), _vp, _vp, _vp]),
    "oea_step_exchange_floats": (_sz, [_i64, _i64, _i32]),
    "oea_step_scratch_elem_bytes": (_i32, []),
    "oea_triple_step_phase": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64,
                                        C.POINTER(StepCfg), _vp, _vp, _i32, _vp]),
    "oea_part_rows_per_rank": (_i64, [_i64, _i32]),
    "oea_part_send_floats": (_sz, [_i64, _i32, _i32]),
    "oea_part_pack": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp]),
    "oea_part_apply": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, C.POINTER(StepCfg), _vp, _i64,
                                 _vp, _vp]),
    "oea_step_normal_scratch": (C.c_int, [_i64, _i64, _i32, _vp, _vp]),
    "oea_step_apply_normals": (C.c_int, [_i64, _i64, _i32, _vp, _vp, _vp]),
    "oea_part_unpack": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "oea_step_scatter_ent_rows": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _i32, _vp]),
    "oea_tripleset_capacity": (_u64, [_i64]),
    "oea_tripleset_build": (C.c_int, [_vp, _i64, _vp, _u64, _vp]),
    "oea_sample_negatives": (C.c_int, [_vp, _i64, _i32, _vp, _u64, _vp, _i32, _vp, _vp, _i32, _u64,
                                       _u32, _u32, _i32, _vp, _vp, _vp]),
    "oea_sample_negatives_replay": (C.c_int, [_vp, _i64, _i32, _vp, _u64, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "oea_tripleset_filter_bits": (C.c_uint64, [_u64]),
    "oea_tripleset_filter_build": (C.c_int, [_vp, _i64, _vp, _u64, _vp]),
    "oea_sample_negatives_pair": (C.c_int, [_vp, _i64, _i64, _i32, C.POINTER(SamplerSide), C.POINTER(SamplerSide),
                                            _u64, _u32, _u32, _i32, _vp, _vp, _vp]),
    "oea_sample_negatives_epoch": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _i32, C.POINTER(SamplerSide),
                                             C.POINTER(SamplerSide), _u64, _u32, _i32, _vp, _vp, _vp]),
    "oea_triple_epoch": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _i32,
                                   C.POINTER(SamplerSide), C.POINTER(SamplerSide), _u64, _u32, _vp, _vp,
                                   C.POINTER(StepCfg), _vp, _vp, _vp, _vp, _vp]),
    "oea_triple_epoch_range": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32,
                                         C.POINTER(SamplerSide), C.POINTER(SamplerSide), _u64, _u32, _vp, _vp,
                                         C.POINTER(StepCfg), _vp, _vp, _vp, _vp, _vp]),
    "oea_triple_epoch_range_shard": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32,
                                         C.POINTER(SamplerSide), C.POINTER(SamplerSide), _u64, _u32, _vp, _vp,
                                         C.POINTER(StepCfg), _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "oea_epoch_layout_bytes": (_sz, [_i64]),
    "oea_epoch_layout": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _u64, _u32, _vp, _vp, _sz, _vp]),
    "oea_step_plan_supported": (_i32, [C.POINTER(StepCfg), _i64, _i64, _i32, _i32]),
    "oea_step_plan_bytes": (_sz, [_i64, _i32, _i64, _i64, _i32]),
    "oea_step_plan_offsets": (C.c_int, [_i64, _i32, _i64, _i64, _i32, _vp]),
    "oea_step_plan_build": (C.c_int, [_vp, _vp, _i32, _vp, _i64, _i32, _i64, _i64, _i32, _vp, _sz, _vp]),
    "oea_triple_epoch_range_plan": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32,
                                        C.POINTER(SamplerSide), C.POINTER(SamplerSide), _u64, _u32, _vp, _vp,
                                        C.POINTER(StepCfg), _vp, _vp, _vp, _vp, _vp, _sz, _i32, _vp]),
    "oea_triple_epoch_range_comm": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32,
                                        C.POINTER(SamplerSide), C.POINTER(SamplerSide), _u64, _u32, _vp, _vp,
                                        C.POINTER(StepCfg), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "oea_rotate_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "oea_rotate_exchange_doubles": (_sz, [_i64, _i64, _i32]),
    "oea_rotate_step": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _i32,
                                  C.POINTER(RotateCfg), _vp, _vp, _i32, _vp]),
    "oea_rotate_lookup": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _i32, _i32, _vp, _i32, _vp]),
    "oea_mapping_workspace_floats": (_sz, [_i64, _i32, _i32]),
    "oea_mapping_step": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _vp, C.c_float, C.c_float, _i32, _vp, _vp,
                                   _vp, _vp, _vp]),
    "oea_mapping_epoch": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _i32, _i64, _vp, _vp, _f32, _f32, _i32,
                                    C.POINTER(StepCfg), _vp, _vp, _vp, _vp, _vp]),
    "oea_step_entity_scratch": (C.c_int, [_vp, _i64, _i64, _i32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "oea_greedy_matching": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "oea_pair_dots": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _vp, _vp, _i64, _vp, _vp]),
    "oea_perm_index": (_u32, [_u32, _u32, _u32]),
    "oea_sample_link_negatives": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _u64,
                                            _u64, _u32, _vp, _vp, _vp, _vp, _u64, _vp]),
    "oea_topk_sym_workspace_bytes": (_sz, [_i64, _i32]),
    "oea_topk_workspace_bytes": (_sz, [_i64, _i64]),
    "oea_topk_inner": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "oea_row_rank_select_f32": (C.c_int, [_vp, _i64, _i32, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _vp]),
    "oea_row_rank_select_f64": (C.c_int, [_vp, _i64, _i32, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _vp]),
    "oea_topk_rows": (C.c_int, [_vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp]),
    "oea_rank_rows": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "oea_rank_workspace_bytes": (_sz, [_i64]),
    "oea_rank_eval": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "oea_quantize_rows_u16": (C.c_int, [_vp, _i64, _i32, _i32, C.c_float, C.c_float, _vp, _i32, _vp]),
    "oea_l1_u16_strip": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp]),
    "oea_rank_l1_grid_rows": (C.c_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i32, _vp, _i32, _i32, _i64, C.c_float, C.c_float, _vp, _vp, _vp, _vp]),
    "oea_pair_l1_f64": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _vp, _i32, _vp, _vp]),
    "oea_rank_l1_grid_rows_csls": (C.c_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i32, _vp, _i32, _i32, _i64, C.c_float, C.c_float, _vp, _vp,
                                             _vp, _vp, _vp, _vp]),
    "oea_pair_l1_sim": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _vp, _i32, _vp, _vp]),
    "oea_rank_metrics": (C.c_int, [_vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp]),
    "oea_rank_eval_bf16_workspace_bytes": (_sz, [_i64, _i32]),
    "oea_rank_eval_bf16": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "oea_rank_eval_bf16_csls": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "oea_sim_bf16_matrix": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _vp, _i64, _vp]),
    "oea_rank_eval_metrics_bf16": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "oea_rank_eval_metrics_workspace_bytes": (_sz, [_i64]),
    "oea_rank_eval_metrics": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "oea_sim_matrix": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _i32, _vp, _i64, _vp]),
    "oea_row_topk_mean": (C.c_int, [_vp, _i64, _i64, _i64, _i32, _vp, _vp]),
    "oea_csls_means_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "oea_csls_means": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "oea_csls_apply": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "oea_spmm_csr": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _vp, _vp, _i32, C.POINTER(CsrSplit), _vp]),
    "oea_sparse_attn_workspace_floats": (_sz, [C.POINTER(AttnGraph)]),
    "oea_sparse_attn_fwd": (C.c_int, [C.POINTER(AttnGraph), _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _i32, _vp]),
    "oea_sparse_attn_bwd": (C.c_int, [C.POINTER(AttnGraph), _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _i32, _vp]),
    "oea_sparse_attn_dz": (C.c_int, [C.POINTER(AttnGraph), _vp, _vp, _vp, _f32, _vp, _vp]),
    "oea_concat_l2n_fwd": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _i32, _vp, _vp, _vp]),
    "oea_concat_l2n_bwd": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _vp, _i32, _vp, _vp, _vp]),
    "oea_pair_loss_l2_fwd": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _i64, _vp, _f32, _f32, _vp, _vp, _vp]),
    "oea_pair_grad_rows": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "oea_align_loss_l1_coef": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "oea_segment_sum_f32": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "oea_pair_rows_build": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "oea_colsum_blocks": (_i32, [_i64]),
    "oea_highway_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "oea_highway_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "oea_gemm_tn_workspace_floats": (C.c_size_t, [_i64, _i32, _i32]),
    "oea_gemm_tn_f32": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i64, _vp, _i32, _vp, _vp]),
    "oea_gemm_tn_plan": (C.c_int, [_i64, _i32, _i32, C.POINTER(_i32), C.POINTER(_i64)]),
    "oea_gemm_tn_partial": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i64, _i32, _i32, _vp, _vp]),
    "oea_gemm_tn_reduce": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _vp]),
    "oea_sigmoid_mix_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "oea_sigmoid_mix_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "oea_relu_axpy_fwd": (C.c_int, [_vp, _vp, C.c_float, _i64, _vp, _vp]),
    "oea_relu_axpy_bwd": (C.c_int, [_vp, _vp, C.c_float, _i64, _vp, _vp]),
    "oea_colsum_prod": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "oea_bias_tanh_fwd": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "oea_bias_tanh_bwd": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "oea_adam_dense": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _i64, _vp]),
    "oea_align_loss_l1": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _vp]),
    "oea_sgd_rows": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _f32, _vp]),
    "oea_gcn_unit_epoch": (C.c_int, [C.POINTER(GcnUnit), _vp, C.POINTER(GcnUnitBuffers), _vp, _vp]),
    "oea_build_unweighted_adj": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp]),
    "oea_build_weighted_adj": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "oea_build_primal_adj": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp]),
    "oea_build_dual_adj": (C.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "oea_build_2hop": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i32, _vp, _i64, _vp, C.POINTER(_i64), _vp]),
    "oea_comm_unique_id": (C.c_int, [_vp]),
    "oea_comm_init": (C.c_int, [_vp, _i32, _i32, C.POINTER(_vp)]),
    "oea_comm_destroy": (C.c_int, [_vp]),
    "oea_comm_rank": (_i32, [_vp]),
    "oea_comm_size": (_i32, [_vp]),
    "oea_allgather_rows": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "oea_comm_reduce_scatter_f32": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "oea_allreduce_f32": (C.c_int, [_vp, _vp, _i64, _vp]),
    "oea_allreduce_f64": (C.c_int, [_vp, _vp, _i64, _vp]),
    "oea_allreduce_i64": (C.c_int, [_vp, _vp, _i64, _vp]),
    "oea_comm_init_callbacks": (C.c_int, [_i32, _i32, _vp, _vp, C.POINTER(_vp)]),
    "oea_comm_allgather": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "oea_comm_reduce_scatter": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "oea_comm_allreduce": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "oea_step_items": (_i64, [C.POINTER(StepCfg), _i64, _i64]),
    "oea_comm_set_alltoallv": (C.c_int, [_vp, _vp]),
    "oea_comm_alltoallv": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "oea_halo_workspace_bytes": (_sz, [_i64, _i32, _i32, _i64, _i32]),
    "oea_halo_buffer_bytes": (_sz, [_i64, _i32, _i64, _i32, _i32]),
    "oea_halo_plan": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _vp, _sz, _vp, _vp, C.POINTER(_i64), _vp]),
    "oea_triple_epoch_range_halo": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32,
                                        C.POINTER(SamplerSide), C.POINTER(SamplerSide), _u64, _u32, _vp, _vp,
                                        C.POINTER(StepCfg), _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "oea_comm_profile_begin": (C.c_int, [_vp]),
    "oea_comm_profile_end": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(_i32)]),
}

# oea_comm_callback: int (*)(void *user, int32 op, const void *send, void *recv, int64 count, int32 dtype, void *stream)
COMM_CALLBACK = C.CFUNCTYPE(C.c_int, _vp, _i32, _vp, _vp, _i64, _i32, _vp)
# oea_comm_alltoallv_callback: int (*)(void *user, const void *send, const int64 *send_counts, const int64 *send_displs, void *recv,
#                                     const int64 *recv_counts, const int64 *recv_displs, int32 dtype, void *stream)
COMM_ALLTOALLV_CALLBACK = C.CFUNCTYPE(C.c_int, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64), _vp, C.POINTER(_i64), C.POINTER(_i64), _i32, _vp)
COMM_ALLREDUCE, COMM_ALLGATHER, COMM_REDUCE_SCATTER = 0, 1, 2
COMM_F32, COMM_F64, COMM_I64 = 0, 1, 2
COMM_PHASES = ("grad", "pack", "reduce_scatter", "apply", "all_gather", "unpack")

def tile_glds():
    """the packed (LDS-DMA) tile path is on unless OEA_TILE_GLDS=0 (csrc/sim_rank.hip)"""
    return os.environ.get("OEA_TILE_GLDS", "1")[:1] != "0"


_lib = None


def load(require_device=True):
    """Load libopenea_hip.so and bind every prototype.  Raises OpenEAHipError when the
    library is absent or (require_device) no HIP device is visible."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OpenEAHipError(
                "libopenea_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (there is no CPU fallback)" % LIB_PATH)
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:
            raise OpenEAHipError("cannot load %s: %s" % (LIB_PATH, e))
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)          # AttributeError = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    if require_device and _lib.oea_device_count() <= 0:
        raise OpenEAHipError("no HIP device visible: the OpenEA hot path runs on MI355X only "
                             "(there is no CPU fallback)")
    return _lib


def check(code):
    if code != 0:
        msg = _lib.oea_last_error().decode("utf-8", "replace") if _lib is not None else ""
        raise OpenEAHipError("libopenea_hip error %d: %s" % (code, msg))
