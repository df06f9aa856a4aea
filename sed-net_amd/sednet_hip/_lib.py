"""ctypes binding of libsedhip.so (include/sednet_hip.h).

The library is the product: there is no CPU fallback. Importing this module without a built
libsedhip.so raises immediately, and every call raises RuntimeError on a non-zero status.
"""
import ctypes
import os

import torch  # noqa: F401  -- FIRST: libsedhip.so must bind to the HIP runtime torch already loaded
              # (torch ships its own libamdhip64; loading /opt/rocm's copy first gives two runtimes
              # and every launch on a torch stream then fails with hipErrorNoDevice)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEDHIP_LIB") or os.path.join(_HERE, "libsedhip.so")      # SEDHIP_LIB: A/B runs of two builds

c_int, c_float, c_size_t, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p
P = c_void_p  # every device pointer / stream travels as void*


class MsOptions(ctypes.Structure):
    """sed_ms_options_t (include/sednet_hip.h): per-call options of the mean-shift iteration entry points."""
    _fields_ = [("schedule", c_int), ("weight_digits", c_int)]


OPT = ctypes.POINTER(MsOptions)

# name -> (restype, argtypes); mirrors include/sednet_hip.h one to one
SIGNATURES = {
    "sed_abi_version": (c_int, []),
    "sed_build_arch": (ctypes.c_char_p, []),
    "sed_pairdist_ms_f32": (c_int, [c_int, c_int, c_int, P, P, c_int, P]),
    "sed_pairdist_knn_f32": (c_int, [c_int, c_int, c_int, c_int, P, P, P, c_int, P]),
    "sed_pairdist_pn_f32": (c_int, [c_int, c_int, c_float, P, P, c_int, P]),
    "sed_row_kth_f32": (c_int, [c_int, c_int, c_int, c_int, P, P, P]),
    "sed_row_topk_idx_f32": (c_int, [c_int, c_int, c_int, c_int, P, P, P]),
    "sed_knn_fused_workspace_bytes": (c_size_t, [c_int, c_int]),
    "sed_knn_fused_max_k": (c_int, []),
    "sed_knn_fused_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, c_size_t, P, P]),
    "sed_knn_pn_fused_f32": (c_int, [c_int, c_int, c_int, c_float, P, P, P, c_size_t, P, P]),
    "sed_knn_fused_order_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_size_t, P, P]),
    "sed_spatial_order_max_points": (c_int, []),
    "sed_spatial_order_f32": (c_int, [c_int, c_int, P, P, P]),
    "sed_knn_fused_far_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, c_size_t, P, P]),
    "sed_csr_spmm_f32": (c_int, [c_int, c_int, c_int, c_size_t, P, P, P, P, c_int, P, c_int, P]),
    "sed_hpnet_affinity_csr_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sed_hpnet_affinity_csr_f32": (c_int, [c_int, c_int, c_int, c_float, P, P, P, P, P, P, P, c_size_t, P]),
    "sed_tsgemm_tn_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "sed_tsgemm_tn_f64": (c_int, [c_int, c_int, c_int, c_int, P, c_int, P, c_int, P, P, c_size_t, P]),
    "sed_ritz_f64": (c_int, [c_int, c_int, c_int, P, P, P, P, P]),
    "sed_lobpcg_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sed_lobpcg_residual_f32": (c_int, [c_int, c_int, c_int, P, P, c_int, P, P, c_size_t, P]),
    "sed_lobpcg_update_f32": (c_int, [c_int, c_int, c_int, c_int, P, P, c_int, P, P]),
    "sed_rank1_add_f32": (c_int, [c_int, c_int, c_int, P, c_int, P, P, c_float, P]),
    "sed_ms_bandwidth_finalize_f32": (c_int, [c_int, c_int, c_float, P, P, P]),
    "sed_ms_iterate_f32": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P]),
    "sed_ms_kth_fused_max_k": (c_int, [c_int]),
    "sed_ms_kth_fused_workspace_bytes": (c_size_t, [c_int, c_int]),
    "sed_ms_kth_fused_f32": (c_int, [c_int, c_int, c_int, c_int, P, P, P, c_size_t, P, c_int, P, P, P]),
    "sed_ms_iterate_workspace_bytes": (c_size_t, [c_int, c_int, c_int, OPT]),
    "sed_ms_iterate_plan": (c_int, [c_int, c_int, c_int, OPT]),
    "sed_ms_iterate_kernel_name": (ctypes.c_char_p, [c_int, c_int, c_int, OPT]),
    "sed_ms_iterate_bounds_f16_kernel_name": (ctypes.c_char_p, [c_int, c_int]),
    "sed_ms_iterate_ws_f32": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, c_size_t, OPT, P]),
    "sed_fps_pivots_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, P]),
    "sed_ms_iterate_bounds_f16_refs": (c_int, [c_int]),
    "sed_ms_iterate_bounds_f16_stats_words": (c_int, []),
    "sed_ms_iterate_bounds_f16_workspace_bytes": (c_size_t, [c_int, c_int]),
    "sed_ms_iterate_bounds_f16_f32": (c_int, [c_int, c_int, c_int, c_int, P, P, P, c_float, P, P, c_float, P, c_size_t, P,
                                              c_int, c_int, c_float, P]),
    "sed_ms_sparse_prepare_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sed_ms_sparse_prepare_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, P, P, P, P, P, P, c_size_t, P]),
    "sed_unsort_rows_f32": (c_int, [c_int, c_int, c_int, P, P, P, P]),
    "sed_ms_nms_workspace_bytes": (c_size_t, [c_int, c_int]),
    "sed_ms_nms_f32": (c_int, [c_int, c_int, c_int, P, P, P, P, P, P, P, P, c_size_t, P, P, P, P]),
    "sed_edgeconv_partials_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sed_edgeconv_fwd_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P, c_float, P, P,
                                     P, c_size_t, c_int, P]),
    "sed_edgeconv_fwd_train_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P, c_float, P,
                                           P, P, P, c_size_t, P]),
    "sed_gn_bwd_partials_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sed_gn_bwd_reduce_f32": (c_int, [c_int, c_int, c_int, c_int, ctypes.c_double, P, c_int, P, c_int, P, P, P, c_int,
                                      c_float, P, P, P, P, P, c_size_t, P]),
    "sed_gn_bwd_apply_f32": (c_int, [c_int, c_int, c_int, c_int, P, P, c_int, P, P]),
    "sed_edgeconv_bwd_partials_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "sed_edgeconv_bwd_edge_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "sed_edgeconv_bwd_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P, P, P, P, P, P,
                                     c_int, P, c_size_t, P, P, P, c_size_t, c_int, P]),
    "sed_pair_entropy_partials": (c_size_t, [c_int]),
    "sed_pair_entropy_f32": (c_int, [c_int, c_int, P, c_int, c_int, c_float, P, P, P]),
    "sed_pair_entropy_split_bytes": (c_size_t, [c_int]),
    "sed_pair_entropy_split_f32": (c_int, [c_int, c_int, P, c_int, P, P]),
    "sed_pair_entropy_mfma_f32": (c_int, [c_int, P, c_int, c_float, P, P, P]),
    "sed_pointwise_partials_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sed_pointwise_colext_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sed_pointwise_fwd_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P, c_int, P, P, c_int, P]),
    "sed_pointwise_split_weights_bytes": (c_size_t, [c_int, c_int]),
    "sed_pointwise_split_weights_f32": (c_int, [c_int, c_int, c_int, P, c_int, P, P]),
    "sed_pointwise_fwd_split_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P, c_int, P, P, c_int, P]),
    "sed_pointwise_fwd_split_gn_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P, c_int, c_int, P, P, P, c_int, P, P,
                                               c_int, P]),
    "sed_segment_metrics_workspace_bytes": (c_size_t, [c_int, c_int]),
    "sed_segment_metrics_f32": (c_int, [c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, P, c_size_t, P]),
    "sed_pointwise_split16_weights_bytes": (c_size_t, [c_int, c_int]),
    "sed_pointwise_split16_weights_f32": (c_int, [c_int, c_int, c_int, P, c_int, P, P]),
    "sed_pointwise_fwd_split16_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P, P, c_int, P, P, c_int, P]),
    "sed_pointwise_fwd_bf16": (c_int, [c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P, c_int, P, P, c_int, P]),
    "sed_edgeconv_fwd_train_bf16": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P, c_float, P,
                                            P, P, P, c_size_t, P]),
    "sed_gemm_splits": (c_int, [c_int, c_int, c_int]),
    "sed_gemm_f32": (c_int, [c_int, c_int, c_int, P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, P, c_size_t, P]),
    "sed_gn_finalize_f32": (c_int, [c_int, c_int, c_int, c_int, ctypes.c_double, c_float, P, P, P]),
    "sed_gn_apply_f32": (c_int, [c_int, c_int, c_int, c_int, P, c_int, P, P, P, c_int, c_float, c_float, P, c_int,
                                 P, c_int, P, P]),
    "sed_gn_apply_fused_f32": (c_int, [c_int, c_int, c_int, P, c_int, P, P, P, c_int, c_int, c_float, P, c_int, P, P, P, c_int, c_int,
                                       P, c_int, c_float, P, c_int, P]),
    "sed_colext_finalize_f32": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, P]),
    "sed_gemv_bias_f32": (c_int, [c_int, c_int, c_int, P, c_int, P, P, P, c_int, P]),
    "sed_log_softmax_f32": (c_int, [c_size_t, c_int, P, c_int, P, c_int, P]),
    "sed_fit_segments_f32": (c_int, [c_int, c_int, c_int, P, P, P, P, P, c_int, c_float, c_int, P, P, P]),
    "sed_residual_segments_f32": (c_int, [c_int, c_int, c_int, P, P, P, P, P, c_int, P, P, P]),
    "sed_lstsq3_f32": (c_int, [c_int, P, P, P, P]),
    "sed_chamfer_fwd_f32": (c_int, [c_int, c_int, c_int, P, P, P, P, P, P, P]),
    "sed_chamfer_bwd_f32": (c_int, [c_int, c_int, c_int, P, P, P, P, P, P, P, P, P]),
    "sed_furthest_point_sampling_f32": (c_int, [c_int, c_int, c_int, P, P, P, P]),
    "sed_ball_query_f32": (c_int, [c_int, c_int, c_int, c_float, c_int, P, P, P, P]),
    "sed_group_points_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, P]),
    "sed_three_nn_f32": (c_int, [c_int, c_int, c_int, P, P, P, P, P]),
    "sed_three_interpolate_f32": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P]),
    "sed_row_normalize_f32": (c_int, [c_size_t, c_int, c_int, P, c_int, P, c_int, P]),
    "sed_row_argmax_f32": (c_int, [c_size_t, c_int, P, c_int, P, P]),
    "sed_segment_type_vote": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (or `make -C sed-net_amd/csrc`). "
            "There is no CPU fallback for the SED-Net HIP path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here == header / library drift
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()

_STATUS = {-1: "SED_EINVAL (bad argument)", -2: "SED_EUNSUPPORTED (size outside the instantiated range)"}


def check(status, what):
    if status != 0:
        raise RuntimeError(f"libsedhip: {what} failed with status {status} {_STATUS.get(status, '(hipError_t)')}")


def ptr(t):
    """device pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("libsedhip needs contiguous tensors")
    if not t.is_cuda:
        raise RuntimeError("libsedhip needs tensors resident on the GPU (no CPU fallback)")
    return c_void_p(t.data_ptr())


def stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
