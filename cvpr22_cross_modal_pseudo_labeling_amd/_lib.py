"""ctypes binding of libovis_hip.so (C ABI: include/ovis_hip.h).  No torch types cross it."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libovis_hip.so")

OVIS_OK = 0
_ERRORS = {-1: "OVIS_EINVAL (bad size or null pointer)",
           -2: "OVIS_ENOSPC (workspace too small)",
           -3: "OVIS_ERANGE (problem size not supported by the kernel)"}

_vp, _i, _f, _sz, _l = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_long

# name -> (restype, argtypes); must list every symbol include/ovis_hip.h declares
SIGNATURES = {
    "ovis_version": (ctypes.c_char_p, []),
    "ovis_roi_align_forward_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "ovis_roi_align_forward_workspace_bytes": (_sz, [_i, _i, _i]),
    "ovis_roi_align_forward_mfma_supported": (_i, [_i, _i, _i, _i]),
    "ovis_roi_align_forward_ws_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "ovis_roi_align_forward_strided_nhwc_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "ovis_roi_align_forward_strided_pair_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "ovis_roi_align_forward_strided_from_nhwc_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "ovis_roi_align_backward_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "ovis_roi_align_backward_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ovis_roi_align_backward_plane_supported": (_i, [_i, _i, _i, _i]),
    "ovis_roi_align_backward_ws_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "ovis_roi_align_backward_strided_ws_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "ovis_roi_align_backward_strided_nhwc_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ovis_roi_align_backward_strided_nhwc_ws_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "ovis_roi_pool_forward_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "ovis_roi_pool_backward_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ovis_deform_psroi_pool_forward_f32": (_i, [_vp] * 5 + [_i] * 7 + [_f] + [_i] * 5 + [_f, _vp]),
    "ovis_deform_psroi_pool_backward_f32": (_i, [_vp] * 7 + [_i] * 7 + [_f] + [_i] * 5 + [_f, _vp]),
    "ovis_nms_workspace_bytes": (_sz, [_i]),
    "ovis_nms_f32": (_i, [_vp, _vp, _i, _f, _i, _vp, _sz, _vp, _vp, _vp]),
    "ovis_nms_grouped_f32": (_i, [_vp, _vp, _vp, _i, _f, _i, _vp, _sz, _vp, _vp, _vp]),
    "ovis_sample_fg_bg": (_i, [_vp, _i, _i, _i, ctypes.c_uint64, _vp, _vp, _vp, _vp]),
    "ovis_project_pasted_masks_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "ovis_nms_presorted_workspace_bytes": (_sz, [_i, _i]),
    "ovis_nms_presorted_batched_f32": (_i, [_vp, _vp, _i, _i, _f, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "ovis_box_decode_f32": (_i, [_vp, _l, _vp, _l, _l, _i, _f, _f, _f, _f, _f, _i, _vp, _vp, _vp, _vp]),
    "ovis_smooth_l1_picked_fwd_bwd_f32": (_i, [_vp, _l, _i, _i, _vp, _l, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp]),
    "ovis_gather_rows": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ovis_rois_from_boxes_f32": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "ovis_deform_conv_implicit_f32": (_i, [_vp, _vp, _vp, _vp, _l, _vp, _vp, _l] + [_i] * 16 + [_vp]),
    "ovis_topk_sorted_workspace_bytes": (_sz, [_i, _i]),
    "ovis_topk_sorted_f32": (_i, [_vp, _l, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "ovis_rpn_decode_f32": (_i, [_vp, _l, _l, _l, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp]),
    "ovis_sigmoid_focal_loss_forward_f32": (_i, [_vp, _vp, _vp, _i, _i, _f, _f, _vp]),
    "ovis_sigmoid_focal_loss_backward_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _f, _vp]),
    "ovis_split_bf16x3_f32": (_i, [_vp, _l, _vp, _l, _i, _i, _vp]),
    "ovis_im2col_split_bf16x3_f32": (_i, [_vp, _vp, _l, _i, _i, _i, _i, _i, _i, _vp]),
    "ovis_bias_act_f32": (_i, [_vp, _vp, _vp, _l, _i, _i, _vp]),
    "ovis_split_pair_f32": (_i, [_vp, _l, _vp, _l, _i, _vp]),
    "ovis_gate_split_pair_rows_f32": (_i, [_vp, _l, _vp, _i, _vp, _vp, _l, _i, _vp, _i, _vp, _vp, _vp]),
    "ovis_gate_split_pair_f32": (_i, [_vp, _l, _vp, _i, _vp, _vp, _l, _i, _vp, _i, _vp]),
    "ovis_im2col_nchw_pair_f32": (_i, [_vp, _vp] + [_i] * 9 + [_vp]),
    "ovis_weight_prep_pair_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ovis_slab_reduce_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ovis_im2col_pair": (_i, [_vp, _vp, _l, _i, _i, _i, _i, _i, _vp]),
    "ovis_split_gemm_tn_slices": (_i, [_l, _i, _i, _i]),
    "ovis_split_gemm_pair_tn": (_i, [_vp, _l, _vp, _l, _vp, _i, _l, _i, _i, _i, _i, _i, _i, _vp]),
    "ovis_split_gemm_pair_gated": (_i, [_vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _l] + [_i] * 8 + [_vp]),
    "ovis_split_gemm_pair_gated_ws": (_i, [_vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _l] + [_i] * 7 + [_vp, _sz, _i, _vp]),
    "ovis_split_gemm_pair_workspace_bytes": (_sz, [_l, _i, _i, _i, _i, _i, _i]),
    "ovis_split_gemm_pair_workspace_bytes_ex": (_sz, [_l, _i, _i, _i, _i, _i, _i, _i]),
    "ovis_split_gemm_pair_rp_gated": (_i, [_vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _l, _i, _i, _i, _vp]),
    "ovis_split_gemm_pair_pool_supported": (_i, [_l, _i, _i, _i]),
    "ovis_split_gemm_pair_rp_pool": (_i, [_vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _vp, _l, _l, _i, _i, _i, _vp, _i, _f, _vp]),
    "ovis_split_gemm_pair_rp": (_i, [_vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _vp, _l, _l, _i, _i, _i, _vp, _sz, _i, _vp]),
    "ovis_split_gemm_pair": (_i, [_vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _vp, _l, _l] + [_i] * 9
                             + [_vp, _sz, _i, _vp]),
    "ovis_match_encode_f32": (_i, [_vp, _vp, _vp, _i, _i, _f, _f, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "ovis_rpn_match_encode_f32": (_i, [_vp, _vp, _vp, _i, _i, _f, _f, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "ovis_project_masks_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ovis_gemm_f32": (_i, [_vp, _l, _l, _vp, _l, _l, _vp, _vp, _l, _i, _i, _i, _vp]),
    "ovis_gemm_ex_f32": (_i, [_vp, _l, _l, _vp, _l, _l, _vp, _i, _f, _i, _vp, _l, _i, _i, _i, _vp]),
    "ovis_gemm_f32_workspace_bytes": (_sz, [_i, _i, _i]),
    "ovis_gemm_ex_ws_f32": (_i, [_vp, _l, _l, _vp, _l, _l, _vp, _i, _f, _i, _vp, _l, _i, _i, _i, _vp, _sz, _vp]),
    "ovis_deform_im2col_f32": (_i, [_vp, _vp, _vp, _vp] + [_i] * 13 + [_vp]),
    "ovis_deform_col2im_f32": (_i, [_vp, _vp, _vp, _vp] + [_i] * 13 + [_vp]),
    "ovis_deform_col2im_coord_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp] + [_i] * 13 + [_vp]),
    "ovis_deform_im2col_pair_rows_f32": (_i, [_vp, _vp, _vp, _vp, _l] + [_i] * 15 + [_vp]),
    "ovis_deform_col2im_rows_f32": (_i, [_vp, _l, _vp, _vp, _vp, _vp, _vp, _vp] + [_i] * 15 + [_vp]),
    "ovis_region_noun_align_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ovis_text_embed_f32": (_i, [_vp, _l, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "ovis_project_polygon_masks_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ovis_sgd_momentum_multi_f32": (_i, [_vp, _vp, _i, _f, _f, _f, _i, _vp]),
    "ovis_bottleneck_identity_backward_workspace_bytes": (_sz, [_l, _i, _i, _i, _i, _i, _i]),
    "ovis_bottleneck_identity_backward": (_i, [_vp, _l] * 7 + [_vp] * 3 + [_l, _i, _i, _i, _i, _i, _i] + [_vp] * 7 + [_sz, _i, _vp, _vp]),
    "ovis_weight_prep_pair_multi_f32": (_i, [_vp, _vp, _i, _i, _vp]),
    "ovis_weight_prep_tile": (_i, []),
    "ovis_sgd_chunk_elements": (_i, []),
    "ovis_weighted_ce_fwd_bwd_f32": (_i, [_vp, _vp, _f, _vp, _vp, _vp, _i, _i, _vp]),
    "ovis_mask_bce_stochastic_fwd_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ovis_mask_bce_stochastic_classes_fwd_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
}

_lib = None


def load():
    """Load the HIP library; fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(make -C cvpr22_cross_modal_pseudo_labeling_amd/csrc). There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc == OVIS_OK:
        return
    if rc < 0:
        raise RuntimeError(f"{what}: {_ERRORS.get(rc, rc)}")
    raise RuntimeError(f"{what}: HIP error {rc}")
