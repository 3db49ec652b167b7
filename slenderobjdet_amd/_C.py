"""ctypes binding of ``libslender_hip.so`` (the C-ABI HIP library declared in ``include/slender_hip.h``).

Plays the role ``slender_det._C`` (pybind module built by the reference's ``setup.py:41-86`` from
``slender_det/layers/csrc/vision.cpp:64-80``) plays in the reference: the single native entry point of the
Python host.  There is deliberately NO fallback: if the library is missing or a call fails, we raise.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# SOD_HIP_LIB: another build of the same library (A/B experiments); the default is the in-tree build
LIB_PATH = os.environ.get("SOD_HIP_LIB") or os.path.join(_HERE, "libslender_hip.so")

_lib = None

_P = c_void_p
_I = c_int
_L = c_longlong
_F = c_float

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/slender_hip.h
_SIGS = {
    "sod_conv2d_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _L, _L, _I, _I, _P],
    "sod_conv2d_dgrad": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _L, _P],
    "sod_conv2d_fwd_bits": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "sod_conv2d_dgrad_bits": [_P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "sod_conv2d_wgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _L, _I, _I, _P, _L, _P],
    "sod_conv2d_wgrad_workspace_bytes": [],
    "sod_conv2d_fwd_ml": [_I, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _I, _I, _P],
    "sod_conv2d_fwd_ml_gnsum": [_I, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _I, _P, _I, _P],
    "sod_groupnorm_apply_ml": [_I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _F, _I, _P],
    "sod_conv2d_dgrad_ml": [_I, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _P],
    "sod_conv2d_dgrad_ml_mask": [_I, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _P],
    "sod_conv2d_dgrad_ml_kpitch": [_I, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _P],
    "sod_conv2d_dgrad_ml_accum": [_I, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _P],
    "sod_conv2d_wgrad_ml": [_I, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _I, _I, _P, _L, _P],
    "sod_groupnorm_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _F, _I, _P, _L, _P],
    "sod_groupnorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _I, _P, _L, _P],
    "sod_groupnorm_fwd_ml": [_I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _F, _I, _P, _L, _P],
    "sod_groupnorm_bwd_ml": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P, _L, _P],
    "sod_relu_fwd": [_P, _P, _L, _P],
    "sod_relu_bwd": [_P, _P, _P, _L, _P],
    "sod_add_bf16": [_P, _P, _P, _L, _P],
    "sod_add_up2_bf16": [_P, _P, _P, _I, _I, _I, _I, _P],
    "sod_bias_grad": [_P, _P, _I, _I, _I, _L, _P, _L, _P],
    "sod_bias_grad_scaled": [_P, _P, _P, _P, _F, _F, _I, _I, _I, _L, _P, _L, _P],
    "sod_bias_grad_ml": [_I, _P, _P, _I, _P, _I, _P],
    "sod_maxpool3x3s2": [_P, _P, _I, _I, _I, _I, _P],
    "sod_upsample2x_bwd": [_P, _P, _I, _I, _I, _I, _P],
    "sod_conv2d_dgrad_cwin": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "sod_conv_set_tile256": [_I],
    "sod_conv_set_pw": [_I],
    "sod_conv_set_ws3": [_I],
    "sod_conv_set_wgrad_variant": [_I],
    "sod_conv_set_reverse": [_I],
    "sod_conv_last_variant": [],
    "sod_conv_prof_enable": [_I],
    "sod_conv_prof_collect": [_P, _P, _P, _P, _I],
    "sod_weight_prep_batched": [_P, _P, _P, _I, _L, _P, _P, _P],
    "sod_weight_prep": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "sod_scale_rows": [_P, _P, _I, _L, _P],
    "sod_sgd_step": [_P, _P, _P, _P, _I, _P, _F, _F, _I, _I, _F, _P],
    "sod_adaptive_step": [_P, _P, _P, _P, _P, _I, _I, _F, _F, _F, _F, _F, _F, _F, _P],
    "sod_preprocess_image": [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P],
    "sod_preprocess_batch": [_I, _P, _I, _I, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "sod_nchw_f32_to_nhwc_bf16": [_P, _P, _I, _I, _I, _P],
    "sod_stem_fused": [_I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P],
    "sod_bottleneck_frozen_fwd": [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "sod_resize_flip_preprocess_batch": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "sod_sigmoid_focal_loss_fwd": [_P, _P, _P, _L, _I, _I, _F, _F, _P, _P, _P, _P],
    "sod_sigmoid_focal_loss_fwd_grad": [_P, _P, _L, _I, _I, _F, _F, _P, _P, _P, _I, _P],
    "sod_sigmoid_focal_loss_bwd": [_P, _P, _P, _L, _I, _I, _F, _F, _P, _P, _F, _F, _P, _I, _I, _P],
    "sod_iou_loss_fwd": [_P, _P, _P, _P, _I, _L, _I, _P, _P, _P, _P],
    "sod_iou_loss_bwd": [_P, _P, _P, _P, _I, _L, _I, _P, _P, _P],
    "sod_fcos_assign": [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _F, _I, _P, _P, _P, _P, _P, _P],
    "sod_fcos_regctr_loss_fwd": [_P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "sod_fcos_regctr_loss_bwd": [_P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _P, _P, _P, _F,
                                 _P, _I, _I, _P, _I, _I, _P, _P, _P],
    "sod_fcos_finalize_losses": [_P, _P, _P, _F, _P, _P],
    "sod_nms_workspace_bytes": [_I],
    "sod_fcos_decode": [_P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P, _P],
    "sod_dense_topk_select": [_P, _I, _I, _I, _P, _I, _I, _F, _I, _P, _P, _P, _P, _P],
    "sod_batched_nms_workspace_bytes": [_I, _I, _I],
    "sod_batched_nms_prepare": [_P, _P, _P, _I, _I, _I, _P, _P],
    "sod_batched_nms_run": [_P, _I, _I, _I, _F, _I, _P, _P, _P, _P],
    "sod_rpn_clip_filter": [_P, _P, _P, _I, _I, _I, _F, _P, _P],
    "sod_nms": [_P, _P, _I, _F, _P, _P, _P, _P],
    "sod_nms_rotated": [_P, _P, _I, _F, _P, _P, _P, _P],
    "sod_box_iou_rotated": [_P, _I, _P, _I, _P, _P],
    "sod_roi_align_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P],
    "sod_roi_align_fwd_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P],
    "sod_roi_align_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P],
    "sod_giou_loss_xyxy": [_P, _P, _L, _F, _P, _P, _P, _P, _P, _P],
    "sod_smooth_l1_loss": [_P, _P, _L, _F, _P, _P, _P, _P, _P, _P],
    "sod_anchor_match": [_P, _I, _P, _I, _F, _F, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "sod_retina_targets": [_P, _I, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P],
    "sod_retina_box_loss_fwd": [_P, _I, _P, _P, _I, _I, _I, _I, _F, _P, _P, _F, _P, _P],
    "sod_retina_giou_loss_fwd": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _F, _P, _P, _F, _P, _P],
    "sod_retina_giou_loss_bwd": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _F, _P, _P, _P, _P],
    "sod_retina_box_loss_bwd": [_P, _I, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "sod_deform_conv_fwd_fused": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "sod_deform_conv_fwd_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "sod_deform_conv_wgrad_fused": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _L, _P],
    "sod_deform_im2col": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "sod_deform_col2im": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "sod_deform_im2col_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "sod_deform_col2im_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "sod_deform_conv_set_window_counter": [_P],
    "sod_deform_conv_set_window_slack": [_I],
    "sod_deform_conv_bwd_fused_supported": [_I, _I, _I, _I, _I, _I, _I],
    "sod_deform_conv_bwd_fused": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "sod_f32_to_bf16": [_P, _P, _L, _P],
    "sod_border_align_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "sod_border_align_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "sod_corner_pool_fwd": [_P, _P, _L, _I, _I, _I, _P],
    "sod_corner_pool_bwd": [_P, _P, _P, _L, _I, _I, _I, _I, _P],
    "sod_anchor_match_rotated": [_P, _I, _P, _I, _F, _F, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "sod_box2box_get_deltas": [_P, _P, _L, _I, _P, _P, _P],
    "sod_box2box_apply_deltas": [_P, _P, _L, _I, _I, _I, _P, _F, _P, _P],
    "sod_bce_logits_loss_fwd": [_P, _P, _L, _P, _P, _P],
    "sod_bce_logits_loss_bwd": [_P, _P, _L, _P, _F, _P, _P],
    "sod_bce_logits_soft_fwd": [_P, _P, _P, _I, _L, _P, _P, _P],
    "sod_bce_logits_soft_bwd": [_P, _P, _P, _I, _L, _P, _F, _P, _P],
    "sod_rpn_loc_loss_fwd": [_P, _P, _P, _L, _I, _F, _P, _P, _P],
    "sod_rpn_loc_loss_bwd": [_P, _P, _P, _L, _I, _F, _P, _F, _P, _P],
    "sod_softmax_ce_fwd": [_P, _P, _I, _I, _I, _P, _P, _P],
    "sod_softmax_ce_bwd": [_P, _P, _I, _I, _I, _P, _F, _P, _P],
    "sod_fastrcnn_box_loss_fwd": [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P],
    "sod_fastrcnn_box_loss_bwd": [_P, _P, _P, _I, _I, _I, _I, _F, _P, _F, _P, _P],
    "sod_reppoints_dcn_offset": [_P, _P, _L, _I, _I, _F, _I, _I, _P],
    "sod_points2bbox_fwd": [_P, _P, _I, _I, _I, _I, _F, _F, _I, _P, _L, _P, _L, _P],
    "sod_points2bbox_bwd": [_P, _L, _P, _L, _I, _I, _I, _I, _F, _I, _P, _P, _P],
    "sod_points2bbox_moment_fwd": [_P, _P, _I, _I, _I, _I, _F, _F, _I, _P, _P, _L, _P],
    "sod_points2bbox_moment_bwd": [_P, _L, _P, _P, _I, _I, _I, _I, _F, _F, _I, _P, _F, _P, _P, _P, _P],
    "sod_reppoints_point_match": [_P, _P, _I, _P, _I, _P, _P, _I, _I, _I, _F, _P, _P, _P],
    "sod_reppoints_labels": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "sod_reppoints_box_loss_fwd": [_P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P],
    "sod_reppoints_box_loss_bwd": [_P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _F, _F, _P, _P],
    "sod_reppoints_finalize": [_P, _P, _P, _P, _F, _I, _F, _P, _P],
    "sod_conv2d_fwd_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _L, _I, _P],
    "sod_conv2d_dgrad_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _I, _P],
    "sod_conv2d_wgrad_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _L, _P],
    "sod_groupnorm_fwd_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "sod_groupnorm_bwd_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "sod_eltwise_f32": [_I, _P, _P, _P, _L, _P],
    "sod_add_up2_f32": [_P, _P, _P, _I, _I, _I, _I, _P],
    "sod_upsample2x_bwd_f32": [_P, _P, _I, _I, _I, _I, _P],
    "sod_maxpool3x3s2_f32": [_P, _P, _I, _I, _I, _I, _P],
    "sod_bias_grad_f32": [_P, _P, _I, _I, _I, _L, _P],
    "sod_preprocess_image_f32": [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P],
    "sod_weight_prep_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "sod_fcos_regctr_loss_bwd_f32": [_P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _P, _P, _P, _F,
                                     _P, _I, _I, _P, _I, _I, _P, _P, _P],
    "sod_sample_labels": [_P, _I, _I, _I, _F, _I, ctypes.c_ulonglong, _P, _P, _P],
    "sod_sample_labels_list": [_P, _I, _I, _I, _F, _I, ctypes.c_ulonglong, _P, _P, _P, _P, _P],
    "sod_compact_samples": [_P, _I, _I, _I, _P, _P, _P],
    "sod_roi_label_batched": [_P, _P, _I, _I, _I, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P],
    "sod_rpn_gather_sampled": [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "sod_rpn_scatter_sampled": [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "sod_retina_box_loss_bwd_f32": [_P, _I, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "sod_retina_giou_loss_bwd_f32": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _F, _P, _P, _P, _P],
    "sod_reduce_workspace_bytes": [],
    "sod_version": [],
    "sod_stream_create_cumask": [_P, _I, _P],
    "sod_stream_destroy": [_P],
    "sod_debug_occupy": [_I, _I, _P],
}
_RESTYPES = {"sod_reduce_workspace_bytes": c_longlong, "sod_conv2d_wgrad_workspace_bytes": c_longlong, "sod_nms_workspace_bytes": c_longlong, "sod_batched_nms_workspace_bytes": c_longlong, "sod_version": c_char_p}


class SlenderHipError(RuntimeError):
    pass


def exported_symbols():
    """Names include/slender_hip.h declares (used by the CPU-side symbol test)."""
    return sorted(_SIGS)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SlenderHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C slenderobjdet_amd/csrc). There is no CPU fallback for the HIP ops."
        )
    # torch ships its own libamdhip64; it must be in the process BEFORE our library is dlopen'ed so that both bind to
    # the same HIP runtime (otherwise our launches see "no device": two runtimes, two contexts)
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    _lib = lib
    return lib


_ERRS = {-1: "SOD_EARG (bad argument / unsupported shape)", -2: "SOD_ESIZE (operand exceeds 2 GiB)", -3: "SOD_EALIGN"}


def call(name, *args):
    """Invoke an int-returning entry point and raise on a non-zero status (mirrors AT_ASSERTM/THCudaCheck ->
    RuntimeError in the reference's extension, BorderAlign_cuda.cu:155-165,202)."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise SlenderHipError(f"{name} failed: {_ERRS.get(rc, 'hipError_t %d' % rc)}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(stream=None):
    import torch

    return ctypes.c_void_p((stream if stream is not None else torch.cuda.current_stream()).cuda_stream)


def reduce_workspace_floats():
    return int(load().sod_reduce_workspace_bytes()) // 4
