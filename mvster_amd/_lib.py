"""ctypes binding of libmvster_hip.so (C ABI: include/mvster_hip.h).

There is deliberately no fallback: if the shared library is missing, importing
`mvster_amd.ops` on a GPU box fails loudly instead of silently running PyTorch ops.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVSTER_LIB") or os.path.join(_HERE, "csrc", "libmvster_hip.so")   # env: A/B builds

ERRORS = {-1: "NULL pointer", -2: "bad shape", -3: "unsupported channel/tile combination", -4: "kernel launch failed"}

_f = ctypes.c_void_p     # device pointers are passed as integers
_i = ctypes.c_int
_l = ctypes.c_long
_fl = ctypes.c_float

SIGNATURES = {
    "mvster_relative_projection": [_f, _f, _i, _i, _f],
    "mvster_relative_projection_multi": [_f, _i, _f, _i, _i, _f],
    "mvster_pack_images": [_f, _i, _f, _i, _i, _i, _f],
    "mvster_forward_prologue": [_f, _i, _f, _i, _i, _i, _f, _i, _f, _f, _i, _f, _i, _i, _i, _i, _f],
    "mvster_warp_agg_fwd": [_f, _f, _f, _f, _f, _f] + [_i] * 9 + [_l] * 3 + [_i, _i, _fl, _i, _f],
    "mvster_warp_agg_fwd_sched": [_f] * 6 + [_i, _f, _f, _f] + [_i] * 9 + [_l] * 3 + [_i, _fl, _i, _f],
    "mvster_warp_agg_bwd": [_f] * 11 + [_i] * 9 + [_l] * 3 + [_i, _i, _fl, _f],
    "mvster_warp_agg_bwd_scratch": [_i] * 8 + [_f, _f],
    "mvster_warp_agg_bwd_sorted_scratch": [_i] * 8 + [_f, _f],
    "mvster_warp_agg_bwd_sorted": [_f] * 11 + [_i] * 9 + [_l] * 3 + [_i, _i, _fl, _f],
    "mvster_init_range": [_f, _i, _f, _i, _i, _i, _i, _i, _f],
    "mvster_schedule_inverse_range": [_f, _f, _f, _i, _i, _i, _i, _f],
    "mvster_schedule_range": [_f, _f, _f, _i, _i, _i, _i, _f],
    "mvster_select_depth": [_f, _f, _f, _f, _i, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _fl, _f],
    "mvster_select_depth_bwd": [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f],
    "mvster_upsample_bilinear": [_f, _f, _i, _i, _i, _i, _i, _f],
    "mvster_upsample_bilinear_multi": [_f, _f, _f, _f, _i, _i, _i, _i, _f],
    "mvster_conv_mfma": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _f, _i, _i, _i, _i, _f],
    "mvster_conv_small": [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f],
    "mvster_conv_narrow": [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f],
    "mvster_conv_narrow4": [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _f],
    "mvster_conv_narrow_pair": [_f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _f],
    "mvster_deconv_select": [_f] * 14 + [_i] * 6 + [_fl, _f],
    "mvster_deconv_small": [_f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _f],
    "mvster_fpn_tail_gather": [_f, _f, _f, _f, _i, _i, _i, _i, _f],
    "mvster_fpn_lateral_up": [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f],
    "mvster_fpn_tail_fused": [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _f],
    "mvster_fpn_tail_gather_bwd": [_f, _f, _i, _i, _i, _i, _i, _f],
    "mvster_pack_conv_weights": [_f, _f] + [_i] * 6 + [_l] * 5 + [_i, _f],
    "mvster_pack_wino_weights": [_f, _f, _i, _i, _i, _i, _l, _l, _l, _l, _l, _i, _f],
    "mvster_pack_wino_batch": [_f, _i, _i, _f],
    "mvster_pack_conv_weights_classes": [_f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f],
    "mvster_conv_wgrad": [_f, _f, _f] + [_i] * 20 + [_f],
    "mvster_conv_wgrad_slots": [_i] * 12,
    "mvster_bn_relu_fwd": [_f, _f, _f, _f, _f, _l, _i, _i, _i, _f],
    "mvster_conv_wgrad_finish": [_f, _f] + [_i] * 10 + [_f],
    "mvster_bn_slots": [_l, _i, _i],
    "mvster_col_sum": [_f, _f, _f, _f, _l, _i, _f],
    "mvster_bn_train_slots": [_l, _i, _i],
    "mvster_bn_train_fwd": [_f] * 10 + [_l, _i, _i, _i, _fl, _fl, _f],
    "mvster_bn_train_bwd": [_f] * 7 + [_l, _i, _i, _i, _f],
    "mvster_bn_fused_ok": [_l, _i, _i, _i],
    "mvster_bn_fwd_fused": [_f] * 11 + [_l, _i, _i, _i, _fl, _fl, _f],
    "mvster_bn_bwd_fused": [_f] * 9 + [_l, _i, _i, _i, _f],
    "mvster_conv_wgrad_finish_batch": [_f, _i, _f],
    "mvster_bn_stats": [_f] * 9 + [_l, _i, _i, _fl, _fl, _f],
    "mvster_bn_relu_bwd_reduce": [_f] * 11 + [_l, _i, _i, _i, _f],
    "mvster_bn_relu_bwd_apply": [_f] * 8 + [_l, _i, _i, _i, _i, _f],
    "mvster_upsample2x_cl_fwd": [_f, _f, _i, _i, _i, _i, _f],
    "mvster_upsample2x_cl_bwd": [_f, _f, _i, _i, _i, _i, _f],
    "mvster_upsample2x_nearest_cl": [_f, _f, _i, _i, _i, _i, _i, _f],
    "mvster_sinkhorn": [_f, _f, _f, _f, _f, _i, _i, _l, _i, _fl, _f],
    "mvster_sinkhorn_continuous": [_f, _f, _f, _f, _f, _f, _i, _i, _l, _i, _fl, _f],
    "mvster_stage_loss_terms": [_f, _f, _f, _f, _f, _f, _i, _i, _l, _i, _f],
    "mvster_stage_loss_slots": [_l],
    "mvster_stage_loss_fwd": [_f] * 9 + [_i, _i, _l, _i, _fl, _fl, _fl, _f],
    "mvster_stage_loss_bwd": [_f] * 6 + [_fl, _fl, _f, _f, _i, _i, _l, _f],
    "mvster_mono_depth_fwd": [_f] * 5 + [_i, _l, _f],
    "mvster_mono_depth_bwd": [_f] * 6 + [_i, _l, _f],
    "mvster_upcat_fwd": [_f, _f, _f, _i, _i, _i, _i, _i, _f],
    "mvster_upcat_bwd": [_f, _f, _f, _i, _i, _i, _i, _i, _f],
    "mvster_fine_weights_fwd": [_f] * 6 + [_i, _i, _i, _f],
    "mvster_fine_weights_bwd": [_f] * 9 + [_i, _i, _i, _f],
    "mvster_fused_adam": [_f, _f, _f, _f, _i, _f, _f, _f, _f] + [ctypes.c_double] * 4 + [_f],
    "mvster_geo_filter": [_f] * 10 + [_i, _i, _i, _fl, _fl, _f],
    "mvster_mfma_probe": [_f, _f, _f, _f],
    "mvster_gather_batch": [_f, _i, _i, _f],
    "mvster_last_kernel": [],
    "mvster_build_flags": [],
}
RESTYPES = {"mvster_last_kernel": ctypes.c_char_p}     # everything else returns an int status

_lib = None


def load():
    """Load the library once; raise with build instructions if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "mvster_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no PyTorch fallback for the HIP path." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        lax = bool(os.environ.get("MVSTER_LIB")) and bool(os.environ.get("MVSTER_LIB_LAX"))   # A/B against an older build
        for name, argtypes in SIGNATURES.items():
            if lax and not hasattr(lib, name):
                continue
            fn = getattr(lib, name)        # AttributeError if the library does not export it
            fn.argtypes = argtypes
            fn.restype = RESTYPES.get(name, ctypes.c_int)
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s (code %d)" % (what, ERRORS.get(rc, "unknown error"), rc))


def has_probes():
    """True when the loaded library is the probe build (experiment switches, kernel forms kept for the record)."""
    lib = load()
    return hasattr(lib, "mvster_build_flags") and bool(lib.mvster_build_flags() & 1)


def last_kernel():
    """Kernel name (profiler spelling) the most recent convolution / fused-warp launch of this thread dispatched."""
    return load().mvster_last_kernel().decode()
