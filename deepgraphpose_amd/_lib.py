"""ctypes binding of the C-ABI in include/dgp_hip.h.  No fallback: if the HIP library is
missing (or was not built) importing the engine raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DGP_HIP_LIB lets scripts/ load a diagnostic build (-DDGP_DIAG) of the same sources
LIB_PATH = os.environ.get("DGP_HIP_LIB") or os.path.join(_HERE, "libdgp_hip.so")


class DgpNetDesc(C.Structure):
    _fields_ = [("depth", C.c_int32), ("num_joints", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32),
                ("max_batch", C.c_int32), ("with_locref", C.c_int32), ("mean_pixel", C.c_float * 3),
                ("bn_eps", C.c_float)]


class DgpTensorView(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 4)]


class DgpLossDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nt", "H", "W", "nj", "nl", "n_visible", "n_hidden", "gm2", "gm3",
                                          "gauss_len", "huber")] + \
               [(n, C.c_float) for n in ("gamma", "lengthscale", "stride", "wn_visible", "wn_hidden",
                                         "locref_loss_weight", "n_frames_total", "n_visible_frames_total")] + \
               [(n, C.c_int32) for n in ("use_wt", "Hin", "Win")] + [("wt_max", C.c_float)]


class DgpConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "N", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "rate", "pad_t", "pad_l", "Ho", "Wo", "relu",
        "res_stride", "res_H", "res_W")]


# every symbol include/dgp_hip.h declares: name -> (restype, argtypes)
_vp, _i32, _f32, _sz = C.c_void_p, C.c_int32, C.c_float, C.c_size_t
SYMBOLS = {
    "dgp_version": (C.c_int, []),
    "dgp_tuning_build": (C.c_int, []),
    "dgp_crc32c": (C.c_uint32, [C.c_void_p, C.c_size_t, C.c_uint32]),
    "dgp_last_error": (C.c_char_p, []),
    "dgp_net_create": (C.c_int, [C.POINTER(DgpNetDesc), C.POINTER(_vp)]),
    "dgp_net_destroy": (None, [_vp]),
    "dgp_net_set_input_size": (C.c_int, [_vp, _i32, _i32]),
    "dgp_dlc_loss_fwd_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, C.c_float, _i32, _vp, _vp, _vp,
                                       _vp, _sz, _vp]),
    "dgp_net_load_weights": (C.c_int, [_vp, C.POINTER(DgpTensorView), _i32]),
    "dgp_net_workspace_bytes": (C.c_int, [_vp, _i32, C.POINTER(_sz)]),
    "dgp_net_output_dims": (C.c_int, [_vp] + [C.POINTER(_i32)] * 4),
    "dgp_net_stats": (C.c_int, [_vp, _i32, C.POINTER(_i32), C.POINTER(C.c_double)]),
    "dgp_forward": (C.c_int, [_vp, _vp, _i32, _vp, _sz, _vp, _vp, _vp, _vp]),
    "dgp_soft_argmax": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "dgp_hard_argmax": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "dgp_pmap_threshold": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    "dgp_infer": (C.c_int, [_vp, _vp, _i32, _vp, _sz, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "dgp_infer_packed": (C.c_int, [_vp, _vp, _i32, _vp, _sz, _f32, _i32, _vp, _vp, _vp]),
    "dgp_net_profile_begin": (C.c_int, [_vp, _i32]),
    "dgp_net_profile_end": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "dgp_net_profile_launch": (C.c_int, [_vp, _i32, C.c_char_p, _i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dgp_loss_scratch_bytes": (C.c_int, [C.POINTER(DgpLossDesc), C.POINTER(_sz)]),
    "dgp_loss_fwd_bwd": (C.c_int, [C.POINTER(DgpLossDesc)] + [_vp] * 18 + [_sz, _vp]),
    "dgp_trainer_create": (C.c_int, [_vp, C.POINTER(_vp)]),
    "dgp_trainer_destroy": (None, [_vp]),
    "dgp_trainer_num_tensors": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "dgp_trainer_tensor_info": (C.c_int, [_vp, _i32, C.c_char_p, _i32, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                          C.POINTER(_i32)]),
    "dgp_trainer_buffer": (_vp, [_vp, _i32]),
    "dgp_trainer_upload": (C.c_int, [_vp, _i32, C.c_int64, _vp, C.c_int64]),
    "dgp_trainer_download": (C.c_int, [_vp, _i32, C.c_int64, _vp, C.c_int64]),
    "dgp_trainer_workspace_bytes": (C.c_int, [_vp, _i32, C.POINTER(_sz)]),
    "dgp_trainer_sync_weights": (C.c_int, [_vp, _vp]),
    "dgp_train_forward": (C.c_int, [_vp, _vp, _i32, _vp, _sz, C.POINTER(_vp), C.POINTER(_vp), _vp]),
    "dgp_train_backward": (C.c_int, [_vp, _i32, _vp, _sz, _vp, _vp, _vp]),
    "dgp_sgd_momentum_clip": (C.c_int, [_vp, _f32, _f32, _f32, C.POINTER(_f32), _vp]),
    "dgp_packed_weight_floats": (_sz, [_i32, _i32, _i32, _i32]),
    "dgp_pack_conv_weights": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp]),
    "dgp_conv2d": (C.c_int, [C.POINTER(DgpConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dgp_conv2d_ranged": (C.c_int, [C.POINTER(DgpConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dgp_tensor_absmax": (C.c_int, [_vp, _sz, _vp, _vp]),
    "dgp_f32_to_h2": (C.c_int, [_vp, _sz, _i32, _vp, _vp]),
    "dgp_h2_to_f32": (C.c_int, [_vp, _sz, _i32, _vp, _vp]),
    "dgp_conv2d_h2": (C.c_int, [C.POINTER(DgpConvDesc), _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    "dgp_net_set_tier": (C.c_int, [_vp, _i32]),
    "dgp_net_get_tier": (C.c_int, [_vp]),
    "dgp_f32_to_h1": (C.c_int, [_vp, _sz, _i32, _vp, _vp]),
    "dgp_h1_to_f32": (C.c_int, [_vp, _sz, _i32, _vp, _vp]),
    "dgp_conv2d_h1": (C.c_int, [C.POINTER(DgpConvDesc), _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    "dgp_chain_h2": (C.c_int, [_i32] * 9 + [_vp, _i32, _vp, _i32] + [_vp] * 6 + [_vp, _i32, _vp, _i32, _vp, _vp, _vp]),
    "dgp_unit_h2": (C.c_int, [_i32] * 7 + [_vp, _i32, _vp, _i32] + [_vp] * 3 + [_i32] + [_vp] * 6 + [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "dgp_chain_h1": (C.c_int, [_i32] * 9 + [_vp, _i32, _vp, _i32] + [_vp] * 6 + [_vp, _i32, _vp, _i32, _vp, _vp, _vp]),
    "dgp_unit_h1": (C.c_int, [_i32] * 7 + [_vp, _i32, _vp, _i32] + [_vp] * 3 + [_i32] + [_vp] * 6 + [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "dgp_net_range_status": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), _vp]),
    "dgp_net_recalibrate": (C.c_int, [_vp]),
    "dgp_net_reset_scales": (C.c_int, [_vp]),
    "dgp_net_copy_scales": (C.c_int, [_vp, _vp, _vp]),
    "dgp_net_widen": (C.c_int, [_vp]),
    "dgp_conv2d_wgrad": (C.c_int, [C.POINTER(DgpConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dgp_conv2d_wgrad_shadow": (C.c_int, [C.POINTER(DgpConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dgp_trainer_set_tier": (C.c_int, [_vp, C.c_int32]),
    "dgp_trainer_get_tier": (C.c_int, [_vp]),
    "dgp_trainer_grad_groups": (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "dgp_trainer_grad_group_wait": (C.c_int, [_vp, C.c_int32, _vp]),
    "dgp_trainer_step_status": (C.c_int, [_vp, _vp, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                               C.POINTER(C.c_int32), _vp]),
    "dgp_trainer_fast_mode": (C.c_int, [_vp, C.c_int32]),
    "dgp_trainer_fast_status": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "dgp_conv2d_dgrad_scratch_bytes": (_sz, [C.POINTER(DgpConvDesc)]),
    "dgp_conv2d_dgrad": (C.c_int, [C.POINTER(DgpConvDesc), _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp]),
    "dgp_motion_energy": (C.c_int, [_vp, C.c_int64, C.c_int32, _vp, _vp, _vp]),
    "dgp_maxpool_3x3s2_same": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dgp_preprocess_u8": (C.c_int, [_vp, C.c_int64, C.POINTER(_f32), _vp, _vp]),
}

_lib = None


class DgpError(RuntimeError):
    pass


def load():
    """Load libdgp_hip.so (once).  Raises if it is absent -- there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DgpError(
            "libdgp_hip.so not found at %s -- build it with `python -m deepgraphpose_amd.build` "
            "(hipcc --offload-arch=gfx950); deepgraphpose_amd has no CPU fallback" % LIB_PATH)
    # PyTorch ships its own libamdhip64; load it FIRST so that libdgp_hip.so binds to the same HIP runtime instance as
    # the tensors and streams it is handed (a second runtime loaded earlier sees "no ROCm-capable device")
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)     # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().dgp_last_error()
        raise DgpError("%s failed (%d): %s" % (what or "dgp call", rc, (msg or b"").decode()))
