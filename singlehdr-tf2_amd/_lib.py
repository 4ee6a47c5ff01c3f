"""ctypes binding of libshdr.so (the C-ABI boundary declared in include/shdr.h).

The product path has NO fallback: if the shared object is missing or a symbol
cannot be resolved, importing/using the ops raises immediately.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SHDR_LIB") or os.path.join(_HERE, "libshdr.so")     # SHDR_LIB: an alternative build (kernel experiments)

c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_f32 = ctypes.c_float
c_ptr = ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    """Mirror of `shdr_conv2d_desc` (include/shdr.h)."""
    _fields_ = [("N", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("C1", ctypes.c_int32), ("C2", ctypes.c_int32),
                ("Cout", ctypes.c_int32), ("KH", ctypes.c_int32), ("KW", ctypes.c_int32),
                ("stride", ctypes.c_int32), ("pad_t", ctypes.c_int32), ("pad_l", ctypes.c_int32),
                ("Ho", ctypes.c_int32), ("Wo", ctypes.c_int32),
                ("x2_scale", ctypes.c_float),
                ("act1", ctypes.c_int32), ("act2", ctypes.c_int32),
                ("res_cstride", ctypes.c_int32), ("y_cstride", ctypes.c_int32),
                ("algo", ctypes.c_int32), ("cout_valid", ctypes.c_int32),
                ("w_batch_stride", ctypes.c_int64),
                ("y_pix_stride", ctypes.c_int32), ("y_off_h", ctypes.c_int32), ("y_off_w", ctypes.c_int32),
                ("y_H", ctypes.c_int32), ("y_W", ctypes.c_int32),
                ("prologue", ctypes.c_int32), ("pool", ctypes.c_int32)]


OP_CONV2D_FWD, OP_CONV2D_DGRAD, OP_CONV2D_WGRAD_WINOGRAD, OP_BATCHNORM, OP_ACT_BWD_BIAS = 0, 1, 2, 3, 4     # shdr_workspace_bytes(op, ...)

# name -> (restype, argtypes); must list every symbol of include/shdr.h
SIGNATURES = {
    "shdr_last_error": (ctypes.c_char_p, []),
    "shdr_version": (ctypes.c_char_p, []),
    "shdr_u8_to_unit_f32": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_ptr]),
    "shdr_resize_cubic_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_pad_symmetric_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_rgbe_encode_f32": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_ptr]),
    "shdr_rgbe_rle_encode": (c_i64, [c_ptr, c_int, c_int, c_ptr, c_i64]),
    "shdr_philox4x32_10": (c_int, [c_ptr, c_ptr, c_ptr]),
    "shdr_camera_expose_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, ctypes.c_uint64, c_ptr]),
    "shdr_jpeg_round_trip_f32": (c_int, [c_ptr] * 6 + [c_int, c_int, c_int, c_ptr]),
    "shdr_flip_rot90_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_f32, c_ptr]),
    "shdr_conv2d_winograd_fused_f32": (c_int, [c_ptr] * 6 + [c_int] * 7 + [c_ptr]),
    "shdr_conv2d_winograd_fused2_f32": (c_int, [c_ptr] * 8 + [c_int] * 8 + [c_ptr]),
    "shdr_conv2d_winograd_fused_up2_f32": (c_int, [c_ptr] * 6 + [c_int] * 7 + [c_ptr]),
    "shdr_conv2d_x3_ok_f32": (c_int, [ctypes.POINTER(ConvDesc)]),
    "shdr_conv2d_x3_filter_elems_f32": (c_i64, [ctypes.POINTER(ConvDesc)]),
    "shdr_conv2d_x3_prepare_filter_f32": (c_int, [ctypes.POINTER(ConvDesc), c_ptr, c_ptr, c_ptr]),
    "shdr_conv2d_fwd_x3_f32": (c_int, [ctypes.POINTER(ConvDesc)] + [c_ptr] * 9),
    "shdr_conv2d_fwd_x3_ranged_f32": (c_int, [ctypes.POINTER(ConvDesc)] + [c_ptr] * 12),
    "shdr_absmax_f32": (c_int, [c_ptr, c_i64, c_ptr, c_ptr]),
    "shdr_x3_split_planes_f32": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "shdr_conv2d_wgrad_x3_ok_f32": (c_int, [ctypes.POINTER(ConvDesc), c_int]),
    "shdr_conv2d_wgrad_x3_f32": (c_int, [ctypes.POINTER(ConvDesc), c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "shdr_config_reload": (None, []),
    "shdr_conv2d_x3_input_absmax_f32": (c_int, [c_ptr, c_i64, c_ptr, c_ptr]),
    "shdr_conv2d_x3n_ok_f32": (c_int, [ctypes.POINTER(ConvDesc)]),
    "shdr_conv2d_x3n_filter_elems_f32": (c_i64, [ctypes.POINTER(ConvDesc)]),
    "shdr_conv2d_x3n_prepare_filter_f32": (c_int, [ctypes.POINTER(ConvDesc), c_ptr, c_ptr, c_ptr]),
    "shdr_conv2d_fwd_x3n_f32": (c_int, [ctypes.POINTER(ConvDesc)] + [c_ptr] * 10),
    "shdr_conv2d_fwd_x3n_ranged_f32": (c_int, [ctypes.POINTER(ConvDesc)] + [c_ptr] * 13),
    "shdr_act_bwd_bias_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr]),
    "shdr_winograd_filter_packed_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_ptr]),
    "shdr_conv2d_wgrad_winograd_f32": (c_int, [c_ptr] * 4 + [c_int] * 7 + [c_f32, c_ptr]),
    "shdr_crc32c": (ctypes.c_uint32, [c_ptr, ctypes.c_uint64, ctypes.c_uint32]),
    "shdr_same_pad": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "shdr_conv2d_fwd_f32": (c_int, [ctypes.POINTER(ConvDesc)] + [c_ptr] * 9),
    "shdr_conv2d_fwd_yrange_f32": (c_int, [ctypes.POINTER(ConvDesc)] + [c_ptr] * 10),
    "shdr_conv2d_plan_f32": (c_int, [ctypes.POINTER(ConvDesc), c_int]),
    "shdr_conv2d_prepared_filter_elems_f32": (c_i64, [ctypes.POINTER(ConvDesc), c_int]),
    "shdr_conv2d_filter_is_plain_f32": (c_int, [ctypes.POINTER(ConvDesc), c_int]),
    "shdr_conv2d_prepare_filter_f32": (c_int, [ctypes.POINTER(ConvDesc), c_int, c_ptr, c_ptr, c_ptr]),
    "shdr_conv2d_workspace_bytes_f32": (c_i64, [ctypes.POINTER(ConvDesc), c_int]),
    "shdr_conv2d_fwd_prepared_f32": (c_int, [ctypes.POINTER(ConvDesc)] + [c_ptr] * 11),
    "shdr_conv2d_fwd_prepared_ranged_f32": (c_int, [ctypes.POINTER(ConvDesc)] + [c_ptr] * 14),
    "shdr_conv2d_projected_ok_f32": (c_int, [ctypes.POINTER(ConvDesc)]),
    "shdr_conv2d_fwd_prepared_projected_f32": (c_int, [ctypes.POINTER(ConvDesc)] + [c_ptr] * 15),
    "shdr_conv2d_fwd_x3_projected_f32": (c_int, [ctypes.POINTER(ConvDesc)] + [c_ptr] * 14),
    "shdr_conv2d_fwd_x3_residual_f32": (c_int, [ctypes.POINTER(ConvDesc)] + [c_ptr] * 12),
    "shdr_conv2d_dgrad_workspace_bytes_f32": (c_i64, [ctypes.POINTER(ConvDesc), c_int]),
    "shdr_conv2d_dgrad_f32": (c_int, [ctypes.POINTER(ConvDesc), c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "shdr_conv2d_dgrad_tracks_range_f32": (c_int, [ctypes.POINTER(ConvDesc), c_int]),
    "shdr_conv2d_dgrad_ranged_f32": (c_int, [ctypes.POINTER(ConvDesc), c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "shdr_workspace_bytes": (c_i64, [c_int, ctypes.POINTER(ConvDesc), c_int]),
    "shdr_soft_hist_bwd_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr]),
    "shdr_conv2d_wgrad_f32": (c_int, [ctypes.POINTER(ConvDesc), c_ptr, c_int, c_ptr, c_ptr, c_ptr]),
    "shdr_filter_transform_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_f32, c_ptr]),
    "shdr_bias_grad_f32": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_ptr]),
    "shdr_soft_hist_fwd_f32": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr]),
    "shdr_lin_frontend_fwd_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_avgpool2_fwd_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_maxpool2_fwd_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_maxpool3s2_fwd_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_resize2x_fwd_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_gap_fwd_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_ptr]),
    "shdr_invcrf_decode_fwd_f32": (c_int, [c_ptr] * 5 + [c_int, c_int, c_int, c_ptr]),
    "shdr_increase_fwd_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_ptr]),
    "shdr_apply_rf_fwd_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_i64, c_int, c_ptr]),
    "shdr_clip_fwd_f32": (c_int, [c_ptr, c_ptr, c_i64, c_f32, c_f32, c_ptr]),
    "shdr_vgg_preprocess_fwd_f32": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_ptr]),
    "shdr_reverse3_fwd_f32": (c_int, [c_ptr, c_ptr, c_i64, c_ptr]),
    "shdr_alpha_blend_fwd_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_ptr]),
    "shdr_pack3_fwd_f32": (c_int, [c_ptr] * 4 + [c_int, c_ptr, c_int, c_i64, c_ptr]),
    "shdr_logc_fwd_f32": (c_int, [c_ptr, c_ptr, c_i64, c_ptr]),
    "shdr_affine_act_f32": (c_int, [c_ptr] * 5 + [c_i64, c_int, c_int, c_ptr]),
    "shdr_pad_channels_f32": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr]),
    "shdr_act_bwd_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr]),
    "shdr_clip_bwd_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_f32, c_ptr]),
    "shdr_add_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    "shdr_add_ranged_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    "shdr_avgpool2_bwd_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_maxpool2_bwd_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_maxpool3s2_bwd_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_resize2x_bwd_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_resize2x_bwd_ranged_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "shdr_gap_bwd_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_ptr]),
    "shdr_upsample_zero2_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_bn_stats_f32": (c_int, [c_ptr] * 6 + [c_i64, c_int, c_f32, c_ptr]),
    "shdr_bn_train_apply_f32": (c_int, [c_ptr] * 6 + [c_i64, c_int, c_f32, c_int, c_ptr]),
    "shdr_bn_train_apply_ranged_f32": (c_int, [c_ptr] * 6 + [c_i64, c_int, c_f32, c_int, c_ptr, c_ptr]),
    "shdr_bn_bwd_f32": (c_int, [c_ptr] * 10 + [c_i64, c_int, c_f32, c_ptr]),
    "shdr_bn_bwd_ranged_f32": (c_int, [c_ptr] * 10 + [c_i64, c_int, c_f32, c_ptr, c_ptr]),
    "shdr_invcrf_decode_bwd_f32": (c_int, [c_ptr] * 7 + [c_int, c_int, c_int, c_ptr]),
    "shdr_increase_bwd_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr]),
    "shdr_apply_rf_bwd_f32": (c_int, [c_ptr] * 5 + [c_int, c_i64, c_int, c_ptr]),
    "shdr_diff_loss_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_i64, c_int, c_ptr]),
    "shdr_diff_loss_bwd_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_i64, c_int, c_int, c_ptr]),
    "shdr_tv_loss_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_tv_loss_bwd_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_logc_bwd_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    "shdr_alpha_mask_f32": (c_int, [c_ptr, c_ptr, c_i64, c_f32, c_ptr]),
    "shdr_alpha_blend_bwd_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    "shdr_vgg_preprocess_bwd_f32": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_ptr]),
    "shdr_winograd_filter_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_ptr]),
    "shdr_winograd_tiles": (c_i64, [c_int, c_int, c_int]),
    "shdr_winograd_input_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_winograd_output_f32": (c_int, [c_ptr] * 5 + [c_int, c_int, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_lin_frontend_bwd_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_alpha_blend_full_bwd_f32": (c_int, [c_ptr] * 5 + [c_i64, c_f32, c_ptr]),
    "shdr_unpack3_f32": (c_int, [c_ptr] * 5 + [c_int, c_int, c_i64, c_ptr]),
    "shdr_sample_dot_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_i64, c_ptr]),
    "shdr_mean_norm_fwd_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_i64, c_f32, c_f32, c_ptr]),
    "shdr_mean_norm_bwd_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_i64, c_f32, c_f32, c_ptr]),
    "shdr_conv2d_packed_filter_elems_f16": (c_i64, [c_int] * 5),
    "shdr_conv2d_pack_filter_f16": (c_int, [c_ptr, c_ptr] + [c_int] * 5 + [c_f32, c_ptr]),
    "shdr_conv2d_fwd_f16": (c_int, [ctypes.POINTER(ConvDesc)] + [c_ptr] * 5 + [c_int, c_ptr]),
    "shdr_conv2d_wgrad_f16": (c_int, [ctypes.POINTER(ConvDesc), c_ptr, c_int, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr]),
    "shdr_conv2d_patch_ok_f16": (c_int, [ctypes.POINTER(ConvDesc)]),
    "shdr_conv2d_fwd_patch_f16": (c_int, [ctypes.POINTER(ConvDesc)] + [c_ptr] * 5 + [c_int, c_ptr]),
    "shdr_conv2d_w3_ok_f16": (c_int, [ctypes.POINTER(ConvDesc)]),
    "shdr_conv2d_fwd_w3_f16": (c_int, [ctypes.POINTER(ConvDesc)] + [c_ptr] * 5 + [c_ptr]),
    "shdr_conv2d_wgrad_alltaps_ok_f16": (c_int, [ctypes.POINTER(ConvDesc), c_int, c_int]),
    "shdr_conv2d_wgrad_alltaps_f16": (c_int, [ctypes.POINTER(ConvDesc), c_ptr, c_int, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr]),
    "shdr_cast_f32_to_f16": (c_int, [c_ptr, c_ptr, c_i64, c_ptr]),
    "shdr_cast_f16_to_f32": (c_int, [c_ptr, c_ptr, c_i64, c_ptr]),
    "shdr_pad_channels_f32_to_f16": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr]),
    "shdr_pack3_f16": (c_int, [c_ptr] * 4 + [c_int, c_ptr, c_int, c_i64, c_int, c_ptr]),
    "shdr_unpack3_f16": (c_int, [c_ptr] * 5 + [c_int, c_int, c_i64, c_int, c_ptr]),
    "shdr_act_bwd_bias_f16": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr]),
    "shdr_add_f16": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr]),
    "shdr_avgpool2_fwd_f16": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_avgpool2_bwd_f16": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_maxpool2_fwd_f16": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_maxpool2_bwd_f16": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_maxpool3s2_fwd_f16": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_maxpool3s2_bwd_f16": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_resize2x_fwd_f16": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_resize2x_bwd_f16": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_upsample_zero2_f16": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_gap_fwd_f16": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_ptr]),
    "shdr_gap_bwd_f16": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_ptr]),
    "shdr_bn_stats_f16": (c_int, [c_ptr] * 6 + [c_i64, c_int, c_f32, c_ptr]),
    "shdr_bn_train_apply_f16": (c_int, [c_ptr] * 6 + [c_i64, c_int, c_f32, c_int, c_ptr]),
    "shdr_bn_bwd_f16": (c_int, [c_ptr] * 10 + [c_i64, c_int, c_f32, c_ptr]),
    "shdr_lin_frontend_fwd_f16": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_lin_frontend_bwd_f16": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "shdr_adam_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_ptr]),
}

_lib = None


def load():
    """Load libshdr.so and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libshdr.so not found at %s -- build it with `python singlehdr-tf2_amd/build.py` "
            "(there is no CPU/PyTorch fallback for the hot path)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().shdr_last_error()
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else ""))
