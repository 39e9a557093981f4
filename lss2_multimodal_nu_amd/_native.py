"""ctypes binding of csrc/liblss_hip.so (the C ABI declared in include/lss_hip.h).

There is NO fallback: if the library is missing or a kernel reports an error the
caller gets an exception.  torch is used only to hand over device pointers and
the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# LSS_HIP_LIB: another build of the same ABI (developer A/B runs of two kernel versions on one box)
LIB_PATH = os.environ.get("LSS_HIP_LIB") or os.path.join(_HERE, "csrc", "liblss_hip.so")

BEV_NCHW_F32, BEV_NHWC_F32, BEV_NHWC_BF16 = 0, 1, 2
DT_F32, DT_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, OUT_F32, OUT_HEAD_MAJOR32, W_RING, W_KS = 0, 1, 2, 16, 32, 64, 128
VALUE_NHWC, VALUE_HEAD_MAJOR = 0, 1

_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol of include/lss_hip.h
SIGNATURES = {
    "lss_abi_version": (_i, []),
    "lss_error_string": (ctypes.c_char_p, [_i]),
    "lss_points_to_voxels": (_i, [_vp] * 7 + [_i] * 8 + [_vp, _vp, _vp, _vp]),
    "lss_geom_to_voxels": (_i, [_vp, _vp, _vp] + [_i] * 5 + [_vp, _vp, _vp]),
    "lss_bucket_points": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "lss_depthnet_softmax_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "lss_camencode_v2_fwd": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp] + [_i] * 7 + [_vp, _vp, _vp]),
    "lss_depth_fuse_softmax_fwd": (_i, [_vp] * 5 + [_i] * 6 + [_vp, _vp]),
    "lss_add_pos_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "lss_deform_attn_fwd": (_i, [_vp, _i] + [_vp] * 4 + [_i] * 7 + [_vp, _vp]),
    "lss_layernorm_fwd": (_i, [_vp, _i, _vp, _vp, ctypes.c_longlong, _i, ctypes.c_float, _vp, _i, _vp]),
    "lss_depthnet_voxels_fwd": (_i, [_vp] * 10 + [_i] * 10 + [_vp] * 5),
    "lss_linear_res_ln_fwd": (_i, [_vp] * 4 + [ctypes.c_longlong, _i, _vp, _vp, _vp, ctypes.c_float, _vp, _vp]),
    "lss_ffn_fused_fwd": (_i, [_vp] * 5 + [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, ctypes.c_float, _vp, _vp]),
    "lss_depthnet_voxels_hostcal_fwd": (_i, [_vp] * 7 + [_i] * 10 + [_vp] * 5),
    "lss_lift_splat_forward_hostcal": (_i, [_vp] * 7 + [_i] * 10 + [_vp] * 8 + [_i, _vp]),
    "lss_lift_splat_fwd": (_i, [_vp] * 3 + [_i] * 9 + [_vp, _i, _vp]),
    "lss_lift_splat_bwd": (_i, [_vp, _i, _vp, _vp, _vp] + [_i] * 9 + [_vp, _vp]),
    "lss_segmented_sum": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "lss_conv2d_packed_weight_bytes": (_sz, [_i] * 5),
    "lss_conv2d_pack_weights": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "lss_conv2d_fwd": (_i, [_vp] * 8 + [_i] * 13 + [_vp]),
    "lss_conv2d_ring_ok": (_i, [_i] * 8),
    "lss_conv2d_ring_packed_weight_bytes": (_sz, [_i, _i]),
    "lss_conv2d_pack_weights_ring": (_i, [_vp, _i, _i, _vp, _vp]),
    "lss_conv2d_pack_weights_ring_dgrad": (_i, [_vp, _i, _i, _vp, _vp]),
    "lss_conv2d_ks_ok": (_i, [_i] * 5),
    "lss_conv2d_ks_packed_weight_bytes": (_sz, [_i, _i]),
    "lss_conv2d_pack_weights_ks": (_i, [_vp, _i, _i, _vp, _vp]),
    "lss_conv2d_pack_weights_ks_dgrad": (_i, [_vp, _i, _i, _vp, _vp]),
    "lss_gather_pack": (_i, [_vp, _i, _vp]),
    "lss_conv_bn_act_train_pack": (_i, [_vp] + [_i] * 8 + [_vp, _vp]),
    "lss_clip_adam_partials": (ctypes.c_longlong, [_vp, _i]),
    "lss_clip_adam_step": (_i, [_vp, _i, _vp, _vp, ctypes.c_longlong] + [ctypes.c_float] * 6 + [_vp]),
    "lss_conv2d_wgrad_timeouts": (_i, []),
    "lss_conv2d_ring_timeouts": (_i, []),
    "lss_conv2d_pack_weights_dgrad": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "lss_conv2d_wgrad_workspace_bytes": (_sz, [_i] * 5),
    "lss_conv2d_wgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
    "lss_conv2d_wgrad4x4_workspace_bytes": (_sz, [_i] * 5),
    "lss_conv2d_s2_dgrad_taps": (_i, [_i, _i]),
    "lss_conv2d_pack_weights_s2_dgrad": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "lss_conv2d_wgrad4x4": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
    "lss_upsample_cat_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "lss_upsample_bwd_nhwc": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "lss_bn_train_workspace_bytes": (_sz, [ctypes.c_longlong, _i]),
    "lss_bn_train_fwd": (_i, [_vp, _vp, ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _i,
                              _vp, _vp, _vp, _vp, _vp]),
    "lss_bn_train_bwd": (_i, [_vp, _vp, _vp, ctypes.c_longlong, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lss_bn_partial_sums": (_i, [_vp] * 5 + [ctypes.c_longlong, _i, _i, _i, _vp, _vp, _vp]),
    "lss_bn_train_fwd_from_sums": (_i, [_vp, _vp, ctypes.c_longlong, _i, _vp, ctypes.c_longlong, _vp, _vp, _vp, _vp,
                                        ctypes.c_float, ctypes.c_float, _i, _vp, _vp, _vp, _vp, _vp]),
    "lss_bn_train_bwd_from_sums": (_i, [_vp, _vp, _vp, ctypes.c_longlong, _i, _vp, ctypes.c_longlong, _vp, _vp, _vp, _i,
                                        _vp, _vp, _vp, _vp, _vp, _vp]),
    "lss_conv_bn_act_train_fwd": (_i, [_vp] * 14 + [_i] * 7 + [ctypes.c_float, ctypes.c_float, _i, _vp]),
    "lss_conv_bn_act_train_bwd": (_i, [_vp] * 11 + [_sz] + [_vp] * 9 + [_i] * 8 + [_vp]),
    "lss_weighted_ce_fwd": (_i, [_vp, _vp, _vp, _i, _i, ctypes.c_longlong, _vp, _vp, _vp, _vp]),
    "lss_weighted_ce_bwd": (_i, [_vp, _vp, _vp, _i, _i, ctypes.c_longlong, _vp, _vp, _vp, _vp]),
    "lss_conv2d_s2d_packed_weight_bytes": (_sz, [_i] * 4),
    "lss_conv2d_pack_weights_s2d": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "lss_conv2d_s2_fwd": (_i, [_vp] * 7 + [_i] * 8 + [_vp]),
    "lss_conv2d_head_fwd": (_i, [_vp] * 8 + [_i] * 9 + [_vp]),
    "lss_conv2d_s2_dual_fwd": (_i, [_vp] * 6 + [_i] * 9 + [_vp]),
    "lss_conv2d_sequence": (_i, [_vp, _i, _vp]),
    "lss_lift_splat_forward": (_i, [_vp] * 10 + [_i] * 10 + [_vp] * 8 + [_i, _i, _vp]),
    "lss_region_pipeline_ok": (_i, [_i] * 9),
    "lss_lift_splat_from_heads": (_i, [_vp] * 9 + [_i] * 9 + [_vp] * 6 + [_i, _vp]),
    "lss_lift_splat_direct_bytes": (_sz, [_i] * 9),
    "lss_lift_splat_forward_desc": (_i, [_vp, _vp]),
    "lss_head_ce_workspace_bytes": (_sz, [_i]),
    "lss_head_ce_fwd": (_i, [_vp] * 5 + [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp]),
    "lss_head_ce_bwd": (_i, [_vp] * 5 + [ctypes.c_longlong, _i, _i] + [_vp] * 7),
    "lss_head1x1_fwd": (_i, [_vp] * 3 + [ctypes.c_longlong, ctypes.c_longlong, _i, _i, _vp, _vp]),
    "lss_head1x1_bwd": (_i, [_vp] * 4 + [ctypes.c_longlong, ctypes.c_longlong, _i, _i] + [_vp] * 5),
    "lss_rccl_unique_id_bytes": (_sz, []),
    "lss_rccl_version": (_i, [_vp]),
    "lss_rccl_get_unique_id": (_i, [_vp]),
    "lss_rccl_comm_init": (_i, [_vp, _i, _i, _vp]),
    "lss_allreduce_bucket": (_i, [_vp, _vp, ctypes.c_longlong, _vp]),
    "lss_rccl_comm_destroy": (_i, [_vp]),
    "lss_nchw_f32_to_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "lss_nhwc_to_nchw_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
}



class ConvLaunch(ctypes.Structure):
    """lss_conv_launch_t of include/lss_hip.h."""
    _fields_ = [(n, _vp) for n in ("x", "x2", "w", "scale", "shift", "residual", "y", "stats",
                                   "head_w", "head_b", "head_out", "y2")] + \
               [(n, ctypes.c_int32) for n in ("B", "H", "W", "Cx", "C2", "up", "Cout", "KH", "KW", "stride", "pad",
                                              "relu", "dt", "head_n", "kind", "split")]


class LiftSplatDesc(ctypes.Structure):
    """lss_lift_splat_desc_t of include/lss_hip.h."""
    _fields_ = [(n, _vp) for n in ("frustum", "inv_post_rots", "post_trans", "combine", "trans", "calib_host", "dx", "bx",
                                   "x", "w", "bias", "voxel", "vox_count", "vox_list", "entries", "cursor",
                                   "direct_entries")] + \
               [("direct_bytes", ctypes.c_ulonglong)] + \
               [(n, _vp) for n in ("depth", "feat", "bev")] + \
               [(n, ctypes.c_int32) for n in ("B", "N", "D", "fH", "fW", "Cin", "C", "X", "Y", "Z", "layout", "math")]


_lib = None


class LssNativeError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LssNativeError(
                "liblss_hip.so is not built: run `python -m lss2_multimodal_nu_amd.build_native` "
                "(there is no CPU / eager fallback for the lift-splat path)")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI is incomplete
            fn.restype, fn.argtypes = res, args
        if L.lss_abi_version() != 1:
            raise LssNativeError("liblss_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().lss_error_string(code).decode()
        exc = ValueError if -100 < code < 0 else LssNativeError  # argument checks vs HIP / RCCL run-time errors
        raise exc("%s failed (%d): %s" % (what, code, msg))


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise LssNativeError("expected a GPU tensor, got %s" % t.device)
    return ctypes.c_void_p(t.data_ptr())


def stream():
    """hipStream_t of torch's current stream on the current device (raw handle: the
    Python-level torch.cuda.current_stream() costs ~9 us per call, this ~0.3 us)."""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def rccl_unique_id():
    """128-byte RCCL unique id (bytes) for `rccl_comm_init`; create on one rank, ship to the others."""
    n = lib().lss_rccl_unique_id_bytes()
    buf = ctypes.create_string_buffer(n)
    check(lib().lss_rccl_get_unique_id(buf), "lss_rccl_get_unique_id")
    return buf.raw


def rccl_comm_init(uid, nranks, rank):
    """Collective: every rank calls it with the same id.  Returns the communicator handle (c_void_p)."""
    comm = ctypes.c_void_p()
    check(lib().lss_rccl_comm_init(ctypes.create_string_buffer(uid, len(uid)), nranks, rank, ctypes.byref(comm)),
          "lss_rccl_comm_init")
    return comm


def rccl_version():
    v = ctypes.c_int()
    check(lib().lss_rccl_version(ctypes.byref(v)), "lss_rccl_version")
    return v.value
