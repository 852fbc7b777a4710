"""ctypes binding of libairpose_hip.so (the C ABI in include/airpose_hip.h).

PyTorch is used for device memory and streams only: every call below hands raw device
pointers and the current HIP stream to the library.  There is NO fallback: if the
shared library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
import threading

import torch  # noqa: F401  (must be imported first so that libamdhip64 is the one torch loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AIRPOSE_HIP_LIB", os.path.join(_HERE, "libairpose_hip.so"))   # override: profiling builds

AP_PREC_FP32, AP_PREC_BF16, AP_PREC_BF16X2, AP_PREC_F16 = 0, 1, 2, 3
AP_ERANGE = -5
AP_RANGE_SLOTS = 8                                          # include/airpose_hip.h
# fp32: exact fp32 MFMA chain | bf16: throughput kernels, bf16 storage | f16: the same kernels with fp16 storage (11 significand
# bits instead of 8 at the same MFMA rate: under the 1e-4 bar; |value| <= 65504) | bf16x2: split-bf16 pairs (fast parity mode)
PRECISIONS = {"fp32": AP_PREC_FP32, "bf16": AP_PREC_BF16, "f16": AP_PREC_F16, "bf16x2": AP_PREC_BF16X2}
HALF_PRECISIONS = ("bf16", "f16")                          # the 16-bit throughput modes (fused layer1 / pair / slab / lean kernels)

_c = ctypes
_vp, _i, _f, _i64p = _c.c_void_p, _c.c_int, _c.c_float, _c.POINTER(_c.c_int64)


class SmplxModelStruct(_c.Structure):
    _fields_ = [("num_verts", _c.c_int32), ("num_joints", _c.c_int32), ("num_faces", _c.c_int32),
                ("num_shape_coeffs", _c.c_int32), ("num_extra", _c.c_int32), ("num_landmarks", _c.c_int32),
                ("v_template", _vp), ("shapedirs", _vp), ("posedirs", _vp), ("J_regressor", _vp),
                ("parents", _vp), ("lbs_weights", _vp), ("faces", _vp), ("extra_joint_verts", _vp),
                ("lmk_faces_idx", _vp), ("lmk_bary_coords", _vp)]


# name -> (restype, argtypes); mirrors include/airpose_hip.h one to one
SIGNATURES = {
    "ap_version": (_c.c_char_p, []),
    "ap_last_error": (_c.c_char_p, []),
    "ap_net_create": (_i, [_c.POINTER(_vp), _i, _i, _i]),
    "ap_net_destroy": (None, [_vp]),
    "ap_net_set_tensor": (_i, [_vp, _c.c_char_p, _vp, _i64p, _i]),
    "ap_net_finalize": (_i, [_vp]),
    "ap_net_precision": (_i, [_vp]),
    "ap_net_set_range_check": (_i, [_vp, _i]),
    "ap_net_range_status": (_i, [_vp, _vp, _i]),
    "ap_net_range_peek": (_i, [_vp]),
    "ap_net_last_conv_launches": (_i, [_vp]),
    "ap_net_range_mark_next": (_i, [_vp, _i]),
    "ap_net_range_slot": (_i, [_vp, _i]),
    "ap_net_parity_probe": (_i, [_vp, _i, _c.c_uint64, _c.POINTER(_c.c_double), _vp]),
    "ap_trunk_fwd": (_i, [_vp, _vp, _i, _vp, _vp]),
    "ap_trunk_fwd_twoview": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "ap_trunk_fwd_twoview_async": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "ap_regressor_fwd": (_i, [_vp] + [_vp] * 6 + [_vp, _i] * 4 + [_i, _i] + [_vp] * 4 + [_vp]),
    "ap_regressor_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "ap_regressor_feat_part": (_i, [_vp, _vp, _i, _vp, _vp]),
    "ap_regressor_step_local": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "ap_regressor_step_finish": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "ap_copenet_fwd": (_i, [_vp] + [_vp] * 6 + [_vp, _i] * 4 + [_i, _i] + [_vp] * 4 + [_vp]),
    "ap_singleview_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ap_singleview_reg": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ap_hmr_reg": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "ap_muhmr_fwd": (_i, [_vp, _vp, _vp] + [_vp, _i] * 6 + [_i, _i] + [_vp] * 4 + [_vp]),
    "ap_hmr_fwd": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "ap_conv2d_nhwc": (_i, [_i] + [_vp] * 6 + [_i] * 9 + [_vp]),
    "ap_set_conv_config": (_i, [_i]),
    "ap_abi_version": (_i, []),
    "ap_debug_set_trace": (_i, [_vp]),
    "ap_net_enable_timing": (_i, [_vp, _i]),
    "ap_net_timing": (_i, [_vp, _c.POINTER(_c.c_double), _i64p, _i]),
    "ap_net_set_chunk": (_i, [_vp, _i]),
    "ap_net_set_dual_stream": (_i, [_vp, _i]),
    "ap_net_set_fold": (_i, [_vp, _i]),
    "ap_net_fold_status": (_i, [_vp, _c.POINTER(_c.c_double)]),
    "ap_net_set_fold_bar": (_i, [_vp, _c.c_double]),
    "ap_net_set_fuse_ief": (_i, [_vp, _i]),
    "ap_net_set_fuse_stem": (_i, [_vp, _i]),
    "ap_net_set_fuse_pool": (_i, [_vp, _i]),
    "ap_net_set_tiled": (_i, [_vp, _i]),
    "ap_net_set_fuse_ds": (_i, [_vp, _i]),
    "ap_net_set_fuse_block": (_i, [_vp, _i]),
    "ap_net_set_fuse_pair": (_i, [_vp, _i]),
    "ap_conv_pair_stream_bytes": (_c.c_int64, [_i, _i, _i]),
    "ap_conv_pair_pack": (_i, [_i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ap_conv_pair_nhwc": (_i, [_i] + [_vp] * 9 + [_i] * 3 + [_vp]),
    "ap_conv_pair_ds_nhwc": (_i, [_i] + [_vp] * 9 + [_i] * 6 + [_vp]),
    "ap_bottleneck64_nhwc": (_i, [_i] + [_vp] * 11 + [_i] * 5 + [_vp]),
    "ap_bottleneck64_tail_nhwc": (_i, [_i] + [_vp] * 15 + [_i] * 4 + [_vp]),
    "ap_block_img_stream_bytes": (_c.c_int64, []),
    "ap_block_img_pack": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "ap_block_img_nhwc": (_i, [_i] + [_vp] * 9 + [_i, _vp]),
    "ap_conv_img3_stream_bytes": (_c.c_int64, []),
    "ap_conv_img3_pack": (_i, [_i, _vp, _vp, _vp]),
    "ap_conv_img3_nhwc": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "ap_conv_s2p_stream_bytes": (_c.c_int64, []),
    "ap_conv_s2p_pack": (_i, [_i, _vp, _vp, _vp]),
    "ap_conv_s2p_nhwc": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "ap_conv_pw_stream_bytes": (_c.c_int64, [_i, _i]),
    "ap_conv_pw_pack": (_i, [_i, _vp, _i, _i, _vp, _vp]),
    "ap_conv_pw_nhwc": (_i, [_i] + [_vp] * 6 + [_i] * 3 + [_vp]),
    "ap_conv_pw_ds_nhwc": (_i, [_i] + [_vp] * 6 + [_i] * 6 + [_vp]),
    "ap_conv_pw_k3s2_nhwc": (_i, [_i] + [_vp] * 5 + [_i] * 4 + [_vp]),
    "ap_net_set_pw_conv": (_i, [_vp, _i]),
    "ap_net_set_fuse_tail": (_i, [_vp, _i]),
    "ap_net_set_even_out": (_i, [_vp, _i]),
    "ap_net_set_img_block": (_i, [_vp, _i]),
    "ap_net_set_img3": (_i, [_vp, _i]),
    "ap_net_set_s2p": (_i, [_vp, _i]),
    "ap_smplx_create": (_i, [_c.POINTER(_vp), _c.POINTER(SmplxModelStruct), _i]),
    "ap_smplx_destroy": (None, [_vp]),
    "ap_smplx_num_joints_out": (_i, [_vp]),
    "ap_smplx_fwd": (_i, [_vp, _i] + [_vp] * 8 + [_vp]),
    "ap_smplx_fwd_fused": (_i, [_vp, _i, _vp, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "ap_smplx_fwd_twoview": (_i, [_vp, _i, _vp, _i, _f, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ap_smplx_set_blend_precision": (_i, [_vp, _i]),
    "ap_smplx_set_fused": (_i, [_vp, _i]),
    "ap_smplx_debug_poison_workspace": (_i, [_vp, _i]),
    "ap_smplx_enable_timing": (_i, [_vp, _i]),
    "ap_smplx_timing": (_i, [_vp, _c.POINTER(_c.c_double), _i64p, _i]),
    "ap_fit_create": (_i, [_c.POINTER(_vp), _vp] + [_vp] * 6 + [_i]),
    "ap_fit_destroy": (None, [_vp]),
    "ap_fit_run": (_i, [_vp, _i] + [_vp] * 8 + [_i, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp]),
    "ap_preprocess_crops": (_i, [_vp, _c.c_int64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ap_rot6d_to_rotmat": (_i, [_vp, _i, _vp, _vp]),
    "ap_rotmat_to_angle_axis": (_i, [_vp, _i, _i, _vp, _vp]),
    "ap_batch_rodrigues": (_i, [_vp, _i, _i, _vp, _vp]),
    "ap_transform_points": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "ap_perspective_projection": (_i, [_vp, _i, _i, _vp, _vp, _f, _f, _vp, _vp, _vp]),
}

ABI_VERSION = 9          # include/airpose_hip.h: AP_ABI_VERSION
_lib = None
_lib_lock = threading.Lock()


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        with _lib_lock:
            if _lib is None:
                if not os.path.isfile(LIB_PATH):
                    raise RuntimeError(
                        "airpose_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
                        "g.build()'` (hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
                L = ctypes.CDLL(LIB_PATH)
                abi = getattr(L, "ap_abi_version", None)
                if abi is None or abi() != ABI_VERSION:      # signatures differ: calling through them would pass pointers as ints
                    raise RuntimeError("airpose_amd: %s exports ABI %s, this binding is written against ABI %d (include/airpose_hip.h: "
                                       "AP_ABI_VERSION) -- rebuild the library" % (LIB_PATH, "?" if abi is None else abi(), ABI_VERSION))
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(L, name)
                    fn.restype, fn.argtypes = res, args
                _lib = L
    return _lib


class RangeError(RuntimeError):
    """AP_ERANGE: a stored activation left the fp16 range (precision="f16"); use precision="bf16" for that checkpoint."""


def check(rc, what):
    if rc != 0:
        msg = lib().ap_last_error().decode("utf-8", "replace")
        raise (RangeError if rc == AP_ERANGE else RuntimeError)("airpose_hip %s failed (status %d): %s" % (what, rc, msg))


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("airpose_amd needs a ROCm GPU (MI355X); torch.cuda.is_available() is False "
                           "and there is no CPU fallback")


def dptr(t, name="tensor"):
    """Raw device pointer of a contiguous float32 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (got %s)" % (name, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (got %s)" % (name, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    return ctypes.c_void_p(t.data_ptr())


def f32c(t, device=None):
    """float32 + contiguous (+ device) view/copy of a tensor: host-side plumbing only."""
    if t is None:
        return None
    if device is not None and t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
