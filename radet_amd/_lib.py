"""ctypes binding of libradet_hip.so (the C ABI declared in include/radet_hip.h).

There is NO fallback: if the shared library is missing or a symbol cannot be resolved the import
of the compute path fails loudly.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C radet_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("RADET_LIB", "libradet_hip.so"))

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_sz = C.c_size_t


class RadetConvDesc(C.Structure):
    """Mirror of `struct RadetConvDesc` (include/radet_hip.h)."""
    _fields_ = [("w", _p), ("bias", _p), ("bn_gamma", _p), ("bn_beta", _p), ("bn_mean", _p), ("bn_var", _p),
                ("wf", _p), ("wft", _p), ("bias_f", _p), ("dwf_slabs", _p), ("dbias_partials", _p),
                ("dw", _p), ("dbias", _p), ("dgamma", _p), ("dbeta", _p),
                ("cout", _i), ("cin", _i), ("kh", _i), ("kw", _i), ("nsplit", _i), ("eps", _f),
                ("wft_ld", _i), ("wft_off", _i), ("w16", _i), ("w_amax", _p), ("wfq", _p), ("w_l1", _p), ("bias_amax", _p),
                ("w_l1t", _p)]


class RadetScales(C.Structure):
    """Mirror of `struct RadetScales` (include/radet_hip.h): amax slots of the fp16 hi / lo arithmetic (device pointers)."""
    _fields_ = [("x_amax", _p), ("w_amax", _p), ("y_amax", _p), ("x1_amax", _p), ("w1_amax", _p), ("y1_amax", _p),
                ("yq", _p), ("yq_amax", _p), ("x_true_amax", _p), ("w_l1", _p), ("bias_amax", _p), ("addend_amax", _p)]


class RadetWgradJob(C.Structure):
    """Mirror of `struct RadetWgradJob` (include/radet_hip.h)."""
    _fields_ = [("dy", _p), ("x", _p), ("slabs", _p), ("dbias_partials", _p), ("gather_table", _p),
                ("M", _i), ("Cin", _i), ("Cout", _i), ("ld_dy", _i), ("KH", _i), ("KW", _i), ("S", _i)]


class RadetTapeOp(C.Structure):
    """Mirror of `struct RadetTapeOp` (include/radet_hip.h, "launch tape")."""
    _fields_ = [("kind", C.c_int32), ("fn", C.c_int32), ("stream", _p), ("event", _p), ("args", C.c_uint64 * 32)]


# name -> (restype, argtypes); must list every function of include/radet_hip.h
SIGNATURES = {
    "radet_gather_table_rows": (_i, [_i]),
    "radet_build_gather_table": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p]),
    "radet_conv2d_igemm": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "radet_conv2d_igemm_pair": (_i, [_p] * 13 + [_i, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "radet_conv2d_igemm_taps": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "radet_conv2d_igemm_classes": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "radet_conv2d_igemm_s": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p, _p]),
    "radet_conv2d_igemm_pair_s": (_i, [_p] * 13 + [_i, _i, _i, _i, _i, _i, _i, _p, _sz, _p, _p]),
    "radet_conv2d_igemm_taps_s": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _sz, _p, _p]),
    "radet_conv2d_igemm_classes_s": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _sz, _p, _p]),
    "radet_conv2d_wgrad_s": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "radet_split_pairs": (_i, [_p, _p, _sz, _i, _i, _p, _p, _p]),
    "radet_merge_pairs": (_i, [_p, _p, _sz, _i, _i, _p, _p]),
    "radet_gn_relu_fwd_q": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p, _i, _p, _p, _p, _p]),
    "radet_gn_relu_fwd_pair_q": (_i, [_p] * 20 + [_i, _i, _i, _f, _i, _p, _i, _p]),
    "radet_gn_relu_bwd_q": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p]),
    "radet_maxpool3x3s2_a": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "radet_maxpool3x3s2_q": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "radet_stem_conv_bn_relu_a": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "radet_upsample_add_a": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "radet_upsample_add_bwd_a": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "radet_relu_bwd_a": (_i, [_p, _p, _p, _p, _sz, _p, _p]),
    "radet_absmax": (_i, [_p, _sz, _p, _p]),
    "radet_amax_slot_words": (_i, []),
    "radet_pred3x3_patch": (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _p, _p, _p, _i, _p]),
    "radet_conv2d_wgrad_splits": (_i, [_i, _i, _i, _i, _i]),
    "radet_conv2d_wgrad": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "radet_conv2d_wgrad_group": (_i, [_p, _i, _i, _p]),
    "radet_fold_weights": (_i, [_p, _i, _p]),
    "radet_unfold_grads": (_i, [_p, _i, _i, _p]),
    "radet_stem_conv_bn_relu": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "radet_maxpool3x3s2": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "radet_maxpool3x3s2_bwd_relu": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "radet_stem_wgrad_splits": (_i, [_i, _i, _i]),
    "radet_stem_wgrad": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "radet_gn_workspace_floats": (_i, [_i, _p, _i]),
    "radet_gn_relu_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p, _i, _p]),
    "radet_gn_relu_fwd_pair": (_i, [_p] * 12 + [_i, _i, _i, _f, _i, _p, _i, _p]),
    "radet_gn_relu_fwd_pair_h": (_i, [_p] * 12 + [_i, _i, _i, _f, _i, _p, _i, _p]),
    "radet_gn_relu_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p]),
    "radet_upsample_add": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "radet_upsample_add_bwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "radet_relu_bwd": (_i, [_p, _p, _p, _p, _sz, _p]),
    "radet_nchw_to_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "radet_nhwc_to_nchw": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "radet_head_loss_ws_ints": (_i, [_i]),
    "radet_head_loss": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _f, _f, _p, _p, _p, _i, _p,
                             _i, _p, _i, _p, _p, _p, _i, _p, _p]),
    "radet_scale_relu": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "radet_decode_ws_bytes": (_sz, [_i, _i, _i]),
    "radet_decode_candidates": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "radet_nms_ws_bytes": (_sz, [_i, _i]),
    "radet_nms": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _f, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "radet_assign_ws_bytes": (_sz, [_i, _i]),
    "radet_assign_points": (_i, [_p, _p, _p, _i, _i, _p, _i, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p]),
    "radet_assign_points_f": (_i, [_p, _p, _p, _i, _i, _p, _i, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p]),
    "radet_resize_linear_u8": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "radet_resize_linear_f": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "radet_gaussian_blur9_u8": (_i, [_p, _p, _p, _p, _p, _i, _i, _p]),
    "radet_sobel_edge": (_i, [_p, _p, _p, _p, _p, _i, _i, _p]),
    "radet_stem_conv_bn_relu_h": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "radet_maxpool3x3s2_h": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "radet_gn_relu_fwd_h": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p, _i, _p]),
    "radet_gn_relu_bwd_h": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p]),
    "radet_upsample_add_h": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "radet_upsample_add_bwd_h": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "radet_relu_bwd_h": (_i, [_p, _p, _p, _p, _sz, _p]),
    "radet_convert_rows": (_i, [_p, _p, _sz, _i, _i, _i, _i, _i, _i, _p]),
    "radet_split_planes": (_i, [_p, _p, _sz, _i, _i, _p]),
    "radet_merge_planes": (_i, [_p, _p, _sz, _i, _i, _p]),
    "radet_gn_relu_fwd_p": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p, _i, _p]),
    "radet_gn_relu_fwd_pair_p": (_i, [_p] * 14 + [_i, _i, _i, _f, _i, _p, _i, _p]),
    "radet_gn_relu_bwd_p": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p]),
    "radet_mbd_ws_bytes": (_sz, [_sz]),
    "radet_mbd": (_i, [_p, _p, _i, _p, _p, _f, _i, _i, _p, _sz, _p, _p]),
    "radet_gdt": (_i, [_p, _p, _i, _p, _p, _p, _p, _p]),
    "radet_mask_max": (_i, [_p, _p, _i, _sz, _p]),
    "radet_mask_transform": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "radet_grid_anchors": (_i, [_p, _p, _i, _i, _p]),
    "radet_bbox_overlaps": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "radet_tblr_encode": (_i, [_p, _p, _p, _i, _p, _i, _p]),
    "radet_tblr_decode": (_i, [_p, _p, _p, _i, _p, _i, _f, _f, _i, _p]),
    "radet_loss_partials": (_i, [_sz]),
    "radet_loss_finalize": (_i, [_p, _i, _p, _f, _p, _p]),
    "radet_sigmoid_focal_loss": (_i, [_p, _p, _p, _i, _sz, _i, _f, _f, _p, _f, _p, _p]),
    "radet_sigmoid_focal_loss_bwd": (_i, [_p, _p, _p, _i, _sz, _i, _f, _f, _p, _p, _p, _f, _p, _p]),
    "radet_bce_logits_loss": (_i, [_p, _p, _p, _i, _sz, _i, _p, _f, _p, _p]),
    "radet_bce_logits_loss_bwd": (_i, [_p, _p, _p, _i, _sz, _i, _p, _p, _p, _f, _p, _p]),
    "radet_giou_loss": (_i, [_p, _p, _p, _sz, _f, _p, _f, _p, _p]),
    "radet_giou_loss_bwd": (_i, [_p, _p, _p, _sz, _f, _p, _p, _p, _f, _p, _p]),
    "radet_threshold_compact": (_i, [_p, _sz, _f, _p, _p, _p]),
    "radet_sqnorm_partials": (_i, [_p, _sz, _p, _i, _p]),
    "radet_adamw_step": (_i, [_p, _p, _p, _p, _sz, _f, _f, _f, _f, _f, _i, _f, _f, _p, _i, _p, _p]),
    "radet_tape_fn_index": (_i, [C.c_char_p]),
    "radet_tape_replay": (_i, [_p, _i, _i, _p]),
    "radet_fill_zero": (_i, [_p, _sz, _p]),
    "radet_copy_d2d": (_i, [_p, _p, _sz, _p]),
    "radet_stream_create_cumask": (_i, [_p, _i, _p]),
}

_lib = None


class RadetHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once). Raises RadetHipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RadetHipError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `make -C radet_amd/csrc` "
            "(or __graft_entry__.build()). radet_amd has no CPU / PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RadetHipError(f"libradet_hip.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, name):
    if rc != 0:
        raise RadetHipError(f"{name} failed with code {rc} (-1 bad argument, -2 launch failure)")


_FN = {}


def call(name, *args):
    fn = _FN.get(name)
    if fn is None:
        if name not in SIGNATURES:             # without argtypes ctypes would pass device pointers as 32-bit ints
            raise RadetHipError(f"{name} has no entry in radet_amd._lib.SIGNATURES")
        fn = _FN[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        raise RadetHipError(f"{name} failed with code {rc} (-1 bad argument, -2 launch failure)")
    if TAPE is not None:
        TAPE.record_call(name, args)


TAPE = None         # a radet_amd.tape.Tape while a step is being recorded (every call above is appended to it)
