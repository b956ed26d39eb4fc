"""ctypes binding of libpinhip.so (the C ABI declared in include/pin_abi.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails, this
module raises.  Build it with ``python -m pin_slam_amd.build`` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (PIN_LIBPINHIP: another build of the same library, for A/B runs of compile-time variants -- scripts/build_variant.sh)
LIB_PATH = os.environ.get("PIN_LIBPINHIP") or os.path.join(HERE, "libpinhip.so")

PIN_MAX_K = 8
PIN_MLP_IN = 11
PIN_FEATURE_DIM = 8
PIN_NONLOCAL = -2
PIN_NBR_QUIRK_BIT = 0x40000000
PIN_GN_NSUMS = 32
PIN_GN_REPLICAS = 16
PIN_ABI_VERSION = 18
PIN_ADAM_ROW_EXCLUDED = -(1 << 31)
PIN_COMM_ID_BYTES = 128

vp = C.c_void_p


class SearchParams(C.Structure):
    _fields_ = [
        ("table", vp), ("pos4", vp), ("cand_off", vp), ("travel_dist", vp), ("global2local", vp),
        ("buffer_size", C.c_int64), ("n_points", C.c_int32), ("n_cand", C.c_int32),
        ("cur_ts", C.c_int32), ("diff_travel_dist_local", C.c_float), ("resolution", C.c_float),
        ("max_valid_dist2", C.c_float),
    ]


class BrickCacheC(C.Structure):
    _fields_ = [
        ("dir_keys", vp), ("dir_vals", vp), ("brick_keys", vp), ("brick_mask", vp), ("brick_base", vp),
        ("entries", vp), ("cand_dx", vp), ("dir_mask", C.c_uint32), ("max_bricks", C.c_int32),
        ("max_entries", C.c_int32), ("n_dilate", C.c_int32), ("dir_pack", vp), ("build_ws", vp), ("build_ws_bytes", C.c_int64),
        ("build_grid", C.c_int32), ("pad_", C.c_int32),
    ]


class Field(C.Structure):
    _fields_ = [
        ("feats", vp), ("certainty", vp), ("orient", vp), ("pos", vp), ("dec", vp),
        ("k", C.c_int32), ("hidden", C.c_int32), ("levels", C.c_int32), ("weighted_first", C.c_int32),
        ("sdf_scale", C.c_float), ("out_dim", C.c_int32), ("dec_image_bytes", C.c_int32), ("dec_image", vp),
    ]


class AdamDense(C.Structure):
    _fields_ = [("param", vp), ("grad", vp), ("exp_avg", vp), ("exp_avg_sq", vp), ("n", C.c_int64), ("image", vp),
                ("hidden", C.c_int32), ("levels", C.c_int32), ("out_dim", C.c_int32),
                ("grad_partial", vp), ("partial_slots", C.c_int32), ("partial_scale", C.c_float)]


class ColorTerm(C.Structure):
    _fields_ = [("field", C.POINTER(Field)), ("colors", vp), ("mode", C.c_int32), ("photo_weight", C.c_float)]


class GnParams(C.Structure):
    _fields_ = [
        ("valid_nn_k", C.c_int32), ("min_grad_norm", C.c_float), ("max_grad_norm", C.c_float),
        ("max_sdf_std", C.c_float), ("gm_dist", C.c_float), ("gm_grad", C.c_float),
        ("dist_div_grad_norm", C.c_int32),
    ]


class GnLoopParams(C.Structure):
    _fields_ = [
        ("lm_lambda", C.c_double), ("term_thre_deg", C.c_double), ("term_thre_m", C.c_double),
        ("min_valid_ratio", C.c_double), ("max_increment_ratio", C.c_double), ("min_valid_points", C.c_int32),
        ("iter_n", C.c_int32), ("early_exit", C.c_int32),
    ]


PIN_GN_STATE_DOUBLES = 80
PIN_GN_STATE_STATUS = 61
PIN_STATUS_FP16_RANGE = 1


class MapArrays(C.Structure):
    _fields_ = [(n, vp) for n in ("table", "pos", "pos4", "orient", "geo", "color", "ts_create", "ts_update",
                                  "certainty")]


class LocalArrays(C.Structure):
    _fields_ = [(n, vp) for n in ("pos", "orient", "geo", "color", "certainty", "ts_update", "global2local")]


class UpdateParams(C.Structure):
    _fields_ = [
        ("travel_dist", vp), ("buffer_size", C.c_int64), ("n_points", C.c_int32), ("capacity", C.c_int32),
        ("n_max", C.c_int32), ("cur_ts", C.c_int32), ("all_new", C.c_int32), ("resolution", C.c_float),
        ("dist2_thre", C.c_float), ("diff_travel_dist_local", C.c_float),
    ]


class LocalParams(C.Structure):
    _fields_ = [
        ("travel_dist", vp), ("n_points", C.c_int32), ("cur_ts", C.c_int32), ("reboot_ts", C.c_int32),
        ("diff_travel_dist_local", C.c_float), ("time_mode", C.c_int32), ("diff_ts_local", C.c_int32),
        ("use_mid_ts", C.c_int32), ("sensor_f64", C.c_int32), ("sensor", C.c_double * 3), ("radius2", C.c_double),
    ]


class RehashParams(C.Structure):
    _fields_ = [("buffer_size", C.c_int64), ("n_points", C.c_int32), ("cur_ts", C.c_int32), ("with_ts", C.c_int32),
                ("use_mid_ts", C.c_int32), ("resolution", C.c_float)]


class PruneParams(C.Structure):
    _fields_ = [("travel_dist", vp), ("n_points", C.c_int32), ("cur_ts", C.c_int32), ("global_prune", C.c_int32),
                ("certainty_thre", C.c_float), ("diff_travel_dist_local", C.c_float)]


class TrainParams(C.Structure):
    _fields_ = [
        ("n_main", C.c_int32), ("n_eik", C.c_int32), ("loss_weight_on", C.c_int32),
        ("sigma", C.c_float), ("weight_e", C.c_float), ("eik_eps", C.c_float),
        ("inv_n_main", C.c_float), ("inv_n_eik", C.c_float), ("eik_analytic", C.c_int32),
        ("dec_image_current", C.c_int32), ("defer_weight_grad", C.c_int32), ("defer_dec_reduce", C.c_int32),
    ]


class TrainGroup(C.Structure):
    _fields_ = [("n_iters", C.c_int32), ("first_step", C.c_int32), ("last_of_call", C.c_int32), ("rows_form", C.c_int32),
                ("query", vp), ("query_stride", C.c_int64), ("nbr", vp), ("nbr_stride", C.c_int64), ("nn", vp), ("nn_stride", C.c_int64),
                ("sdf_label", vp), ("label_stride", C.c_int64), ("sample_weight", vp), ("weight_stride", C.c_int64),
                ("sample_ts", vp), ("ts_stride", C.c_int64), ("certainty_rw", vp), ("ts_update_rw", vp),
                ("feat_grad", vp), ("dec_grad", vp), ("loss_out", vp), ("workspace", vp), ("workspace_bytes", C.c_int64),
                ("n_records", C.c_int64), ("exp_avg", vp), ("exp_avg_sq", vp), ("pending", vp), ("row_flags", vp), ("n_rows", C.c_int64),
                ("coef", vp), ("t_max", C.c_int32), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("dense", AdamDense), ("partial", vp), ("partial_slots", C.c_int32), ("partial_scale", C.c_float),
                ("side_stream", vp), ("fc", vp), ("cp", vp), ("color_label", vp), ("color_stride", C.c_int64),
                ("c_feat_grad", vp), ("c_dec_grad", vp), ("c_loss_out", vp), ("c_workspace", vp), ("c_workspace_bytes", C.c_int64),
                ("c_exp_avg", vp), ("c_exp_avg_sq", vp), ("c_pending", vp), ("c_row_flags", vp), ("c_dense", AdamDense)]


class TrainColorParams(C.Structure):
    _fields_ = [("n_main", C.c_int32), ("loss_weight_on", C.c_int32), ("surface_range", C.c_float),
                ("weight_i", C.c_float), ("dec_image_current", C.c_int32), ("n_main_global", C.c_int32), ("surface_count", vp)]


class SemParams(C.Structure):
    _fields_ = [("n_main", C.c_int32), ("heads", C.c_int32), ("weight_s", C.c_float), ("reserved", C.c_int32), ("labels", vp),
                ("selected", vp), ("count", vp)]


class PreprocessParams(C.Structure):
    _fields_ = [("train_vox", C.c_float), ("source_vox", C.c_float), ("min_z", C.c_float), ("max_z", C.c_float),
                ("min_range", C.c_float), ("max_range", C.c_float), ("correct_deg", C.c_double), ("want_source", C.c_int32),
                ("deskew", C.c_int32), ("pose", C.c_double * 16), ("ts_mid_pose", C.c_double)]


class DpRegions(C.Structure):
    _fields_ = [("boxes", vp), ("world", C.c_int32), ("rank", C.c_int32), ("reach", C.c_int32), ("resolution", C.c_float)]


class PoolArrays(C.Structure):
    _fields_ = [("coord", vp), ("global_coord", vp), ("sdf_label", vp), ("weight", vp), ("ts", vp), ("color", vp),
                ("color_channels", C.c_int32), ("reserved", C.c_int32), ("sem_label", vp)]


class SampleParams(C.Structure):
    _fields_ = [
        ("surface_n", C.c_int32), ("front_n", C.c_int32), ("behind_n", C.c_int32), ("dist_weight_on", C.c_int32),
        ("behind_dropoff_on", C.c_int32), ("frame_id", C.c_int32), ("surface_range", C.c_double),
        ("free_begin_ratio", C.c_double), ("free_end_dist", C.c_double), ("dist_weight_scale", C.c_double),
        ("max_range", C.c_double), ("pose", C.c_double * 12), ("sem_labels", vp),
    ]


i32, i64, f32, f64 = C.c_int32, C.c_int64, C.c_float, C.c_double
P = C.POINTER

# name -> (restype, argtypes).  Every symbol declared in include/pin_abi.h is listed here;
# tests/test_abi_and_dropin_surface.py checks the header, this table and the built library agree.
SIGNATURES = {
    "pin_version": (i32, []),
    "pin_warmup": (i32, []),
    "pin_last_error": (C.c_char_p, []),
    "pin_status": (i32, [vp, i32, vp]),
    "pin_sem_workspace_bytes": (i64, [i32, i32, i32, i32]),
    "pin_sem_select": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "pin_gather_labels_drawn": (i32, [vp, vp, i32, vp, vp, i32, i32, i64, i64, vp, vp]),
    "pin_train_sem_step": (i32, [P(Field), P(SemParams), vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "pin_sem_query": (i32, [P(Field), vp, vp, vp, i32, i32, vp, vp, vp]),
    "pin_decoder_sem": (i32, [P(Field), vp, i32, i32, i32, vp, vp]),
    "pin_candidate_offsets": (i32, [vp, i32, i64, vp]),
    "pin_pack_positions": (i32, [vp, vp, i32, i32, vp, vp]),
    "pin_radius_search": (i32, [P(SearchParams), vp, i32, vp, vp, vp]),
    "pin_knn_query": (i32, [P(SearchParams), vp, i32, i32, vp, vp, vp, vp, vp]),
    "pin_gn_state_init": (i32, [vp, vp, i32, vp]),
    "pin_gn_loop_init": (i32, [vp, vp, i32, vp, vp]),
    "pin_decoder_image_bytes": (i64, [i32, i32]),
    "pin_stage_decoder": (i32, [P(Field), vp, i64, vp]),
    "pin_gn_knn": (i32, [P(SearchParams), P(BrickCacheC), vp, i32, i32, vp, vp, vp, vp, vp]),
    "pin_knn_list_stride": (i32, [i32]),
    "pin_gn_knn_listed": (i32, [P(SearchParams), P(BrickCacheC), vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp]),
    "pin_gn_knn_coherent": (i32, [P(SearchParams), P(BrickCacheC), vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp]),
    "pin_gn_accumulate_dev": (i32, [P(Field), P(GnParams), P(ColorTerm), vp, vp, vp, vp, i32, vp, vp, vp]),
    "pin_gn_solve": (i32, [vp, vp, P(GnLoopParams), vp]),
    "pin_gn_accumulate_solve": (i32, [P(Field), P(GnParams), P(ColorTerm), P(GnLoopParams), vp, vp, vp, vp, i32, vp, vp, vp]),
    "pin_brick_build_workspace_bytes": (i64, [i32, i32]),
    "pin_brick_build": (i32, [P(SearchParams), P(BrickCacheC), vp, vp]),
    "pin_knn_query_bricks": (i32, [P(SearchParams), P(BrickCacheC), vp, i32, i32, vp, vp, vp, vp, vp]),
    "pin_query_feature": (i32, [P(Field), vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp]),
    "pin_decoder_sdf": (i32, [P(Field), vp, i32, vp, vp]),
    "pin_decoder_color": (i32, [P(Field), vp, i32, vp, vp]),
    "pin_color_query": (i32, [P(Field), vp, vp, vp, i32, vp, vp, vp, vp, vp]),
    "pin_sdf_query": (i32, [P(Field), vp, vp, vp, i32, vp, vp, vp, vp, vp]),
    "pin_gn_accumulate": (i32, [P(Field), P(GnParams), P(ColorTerm), vp, vp, vp, vp, i32, vp, vp, vp, vp]),
    "pin_maint_workspace_bytes": (i64, [i32]),
    "pin_spatial_sort": (i32, [vp, i32, f32, vp, vp, vp, i64, vp]),
    "pin_voxel_downsample": (i32, [vp, i32, f32, vp, vp, vp, i64, vp]),
    "pin_voxel_downsample_fast": (i32, [vp, i32, f32, vp, vp, vp, i64, vp]),
    "pin_map_update": (i32, [P(MapArrays), P(UpdateParams), vp, vp, vp, vp, vp, i64, vp]),
    "pin_reset_local_map": (i32, [P(MapArrays), P(LocalArrays), P(LocalParams), vp, vp, vp, i64, vp]),
    "pin_assign_local_to_global": (i32, [P(MapArrays), P(LocalArrays), i32, i32, vp, vp]),
    "pin_gather_batch": (i32, [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]),
    "pin_gather_batch_drawn": (i32, [vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32,
                                     f32, vp]),
    "pin_gather_batches_drawn": (i32, [vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32,
                                       f32, i32, i64, i64, vp]),
    "pin_gather_records_drawn": (i32, [vp, vp, i32, vp, i32, vp, vp, i32, i64, i32, i64, i64, vp, vp, vp]),
    "pin_train_make_queries": (i32, [vp, i32, i32, i32, i32, f32, vp, vp]),
    "pin_train_workspace_bytes": (i64, [i32, i32, i32, i32]),
    "pin_train_step": (i32, [P(Field), P(TrainParams), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "pin_train_weight_grad": (i32, [P(Field), P(TrainParams), vp, vp, vp, i64, vp]),
    "pin_train_deferred_partial": (i32, [vp, vp, vp, vp]),
    "pin_train_group_steps": (i32, [P(Field), P(TrainParams), P(TrainGroup), vp]),
    "pin_train_color_step": (i32, [P(Field), P(TrainColorParams), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "pin_adam_step": (i32, [vp, vp, vp, vp, i64, i32, f32, f32, f32, f32, i32, vp]),
    "pin_mark_rows": (i32, [vp, i64, vp, vp]),
    "pin_adam_step_rows": (i32, [vp, vp, vp, vp, i64, i32, vp, i32, f32, f32, f32, f32, i32, vp]),
    "pin_adam_lazy_prepare": (i32, [vp, i64, vp, vp, vp, vp, vp, i32, vp, i32, f32, f32, f32, P(AdamDense), vp]),
    "pin_adam_lazy_prepare_rows": (i32, [vp, i64, vp, vp, vp, vp, vp, vp, i64, i32, vp, i32, f32, f32, f32, P(AdamDense), vp]),
    "pin_adam_lazy_flush": (i32, [vp, vp, vp, vp, vp, i64, i32, vp, i32, f32, f32, f32, P(AdamDense), vp]),
    "pin_pool_workspace_bytes": (i64, [i64]),
    "pin_sample_rays": (i32, [P(SampleParams), vp, vp, i32, i32, vp, vp, vp, P(PoolArrays), vp]),
    "pin_pool_window_mask": (i32, [vp, i32, vp, f64, vp, vp, vp, vp, i64, vp]),
    "pin_pool_discard": (i32, [vp, vp, vp, i32, vp]),
    "pin_pool_compact": (i32, [P(PoolArrays), P(PoolArrays), vp, i32, i32, vp, vp, i64, vp]),
    "pin_query_certainty": (i32, [P(SearchParams), vp, vp, i32, vp, vp]),
    "pin_new_sample_index": (i32, [vp, vp, i32, f32, f32, i64, vp, vp, vp, i64, vp]),
    "pin_select_surface_points": (i32, [vp, vp, i32, f32, vp, vp, vp, i64, vp]),
    "pin_transform_points": (i32, [vp, i32, i32, vp, vp, vp]),
    "pin_crop_frame": (i32, [vp, i32, i32, vp, f32, f32, f32, f32, vp, vp, vp, vp, i64, vp]),
    "pin_intrinsic_correct": (i32, [vp, i32, i32, f64, vp]),
    "pin_deskew": (i32, [vp, i32, i32, vp, vp, f64, vp, i64, vp]),
    "pin_transform_by_frame": (i32, [vp, i32, vp, vp, i32, vp, vp, vp]),
    "pin_gather_rows": (i32, [vp, i32, vp, i32, vp, vp]),
    "pin_hash_rebuild": (i32, [P(MapArrays), P(MapArrays), P(RehashParams), vp, vp, vp, i64, vp]),
    "pin_prune_map": (i32, [P(MapArrays), P(MapArrays), P(PruneParams), vp, vp, i64, vp]),
    "pin_comm_load": (i32, [C.c_char_p]),
    "pin_comm_unique_id": (i32, [vp]),
    "pin_comm_init_rank": (i32, [vp, i32, i32, P(vp)]),
    "pin_comm_destroy": (i32, [vp]),
    "pin_allreduce_grads": (i32, [vp, vp, i64, vp]),
    "pin_allreduce_f32": (i32, [vp, vp, vp, i64, vp]),
    "pin_dp_kd_boxes": (i32, [vp, i32, i32, vp]),
    "pin_dp_boxes_decode": (i32, [vp, i32, vp, vp]),
    "pin_preprocess_workspace_bytes": (i64, [i32, i32]),
    "pin_preprocess_frame": (i32, [P(PreprocessParams), vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "pin_dp_signature": (i32, [vp, vp, vp, i32, vp, i32, vp, i32, vp, vp]),
    "pin_dp_sample_cells": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, f32, vp, vp]),
    "pin_dp_partition": (i32, [P(DpRegions), vp, vp, i32, vp, vp, i32, i32, i32, i64, i64, vp, i32, vp, i32, vp, i64, vp, vp, f32, vp, vp]),
    "pin_dp_gather": (i32, [vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, i64, i64, vp, i32, vp, i32, vp, i32, vp, vp, vp, vp,
                            vp, vp, f32, vp]),
    "pin_count_draws": (i32, [vp, i64, vp, vp, i64, vp, vp]),
    "pin_certainty_from_records": (i32, [vp, vp, i32, vp, vp, vp, i64, vp, vp, vp]),
    "pin_dp_own_pool": (i32, [vp, i32, i32, vp, vp, i32, vp, vp, vp, i64, vp]),
    "pin_dp_gather_records": (i32, [vp, vp, i32, vp, vp, i32, vp, vp, i64, i64, vp, i32, i32, vp, i32, vp, vp, vp]),
    "pin_dp_mark_halo": (i32, [P(DpRegions), vp, i32, vp, i32, vp, vp, vp, vp, i64, vp]),
    "pin_dp_halo_pack": (i32, [vp, i32, vp, vp, vp]),
    "pin_dp_halo_adam": (i32, [vp, i32, vp, vp, vp, vp, i32, vp, i32, f32, f32, f32, vp]),
    "pin_dp_owner_pack": (i32, [vp, i32, vp, i32, vp, vp]),
    "pin_dp_owner_lists_workspace_bytes": (i64, [i32, i32]),
    "pin_dp_owner_lists": (i32, [vp, i32, i32, vp, vp, vp, i64, vp]),
    "pin_dp_rows_pack": (i32, [vp, i32, vp, vp, vp, vp, vp, vp]),
    "pin_dp_rows_unpack": (i32, [vp, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    "pin_dp_exclude_rows": (i32, [vp, i32, vp, vp]),
    "pin_dp_halo_side_gather": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp]),
    "pin_dp_halo_side_scatter": (i32, [vp, i32, vp, vp, vp, vp, vp]),
    "pin_allgather_f32": (i32, [vp, vp, vp, i64, vp]),
    "pin_dp_cert_snapshot": (i32, [vp, vp, i32, vp]),
    "pin_dp_cert_delta": (i32, [vp, vp, vp, i32, vp]),
    "pin_dp_cert_apply": (i32, [vp, vp, vp, i32, vp]),
    "pin_dp_sync_side_effects": (i32, [vp, vp, vp, vp, vp, i32, vp]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is required (no CPU fallback). "
                "Build it with `python -m pin_slam_amd.build`.")
        # torch bundles its own libamdhip64; it must be the HIP runtime of the process (device
        # pointers and streams come from torch), so make sure it is loaded before our library
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().pin_last_error()
        raise RuntimeError(f"libpinhip {what} failed ({rc}): {msg.decode() if msg else '?'}")
