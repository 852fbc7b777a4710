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
# the fp16 flavour: the same sources built with -DAP_F16 (fp16 instead of bf16 as the 16-bit storage / MFMA type)
LIB_PATH_F16 = os.environ.get("AIRPOSE_HIP_LIB_F16", os.path.join(_HERE, "libairpose_hip_f16.so"))

AP_PREC_FP32, AP_PREC_BF16, AP_PREC_BF16X2 = 0, 1, 2
# fp32: exact fp32 MFMA chain | bf16: throughput mode, bf16 storage | f16: throughput mode, fp16 storage (the kernels of the bf16
# mode from the fp16 flavour of the library: 11 significand bits instead of 8 at the same MFMA rate) | bf16x2: split-bf16 pairs
PRECISIONS = {"fp32": AP_PREC_FP32, "bf16": AP_PREC_BF16, "f16": AP_PREC_BF16, "bf16x2": AP_PREC_BF16X2}
FLAVOUR = {"f16": "f16"}                                    # precision -> library flavour ("" = libairpose_hip.so)

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
    "ap_trunk_fwd": (_i, [_vp, _vp, _i, _vp, _vp]),
    "ap_regressor_fwd": (_i, [_vp] + [_vp] * 6 + [_vp, _i] * 4 + [_i, _i] + [_vp] * 4 + [_vp]),
    "ap_regressor_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "ap_copenet_fwd": (_i, [_vp] + [_vp] * 6 + [_vp, _i] * 4 + [_i, _i] + [_vp] * 4 + [_vp]),
    "ap_singleview_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ap_singleview_reg": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ap_hmr_reg": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "ap_muhmr_fwd": (_i, [_vp, _vp, _vp] + [_vp, _i] * 6 + [_i, _i] + [_vp] * 4 + [_vp]),
    "ap_hmr_fwd": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "ap_conv2d_nhwc": (_i, [_i] + [_vp] * 6 + [_i] * 9 + [_vp]),
    "ap_set_conv_config": (_i, [_i]),
    "ap_debug_set_trace": (_i, [_vp]),
    "ap_net_enable_timing": (_i, [_vp, _i]),
    "ap_net_timing": (_i, [_vp, _c.POINTER(_c.c_double), _i64p, _i]),
    "ap_net_set_chunk": (_i, [_vp, _i]),
    "ap_net_set_dual_stream": (_i, [_vp, _i]),
    "ap_net_set_fold": (_i, [_vp, _i]),
    "ap_net_set_fuse_ief": (_i, [_vp, _i]),
    "ap_net_set_fuse_stem": (_i, [_vp, _i]),
    "ap_net_set_fuse_ds": (_i, [_vp, _i]),
    "ap_net_set_fuse_block": (_i, [_vp, _i]),
    "ap_set_bottleneck_cut": (_i, [_i]),
    "ap_net_set_fuse_pair": (_i, [_vp, _i]),
    "ap_conv_pair_nhwc": (_i, [_vp] * 10 + [_i] * 3 + [_vp]),
    "ap_conv_pair_ds_nhwc": (_i, [_vp] * 9 + [_i] * 6 + [_vp]),
    "ap_bottleneck64_nhwc": (_i, [_vp] * 11 + [_i] * 5 + [_vp]),
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

_libs = {}
_lib_lock = threading.Lock()


def lib(flavour=""):
    """Load (once) and return the shared library of that flavour ("" = bf16 storage, "f16" = fp16 storage); raises if it has
    not been built."""
    L = _libs.get(flavour)
    if L is None:
        with _lib_lock:
            L = _libs.get(flavour)
            if L is None:
                path = LIB_PATH_F16 if flavour == "f16" else LIB_PATH
                if not os.path.isfile(path):
                    raise RuntimeError(
                        "airpose_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
                        "g.build()'` (hipcc, gfx950).  There is no CPU fallback." % path)
                L = ctypes.CDLL(path)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(L, name)
                    fn.restype, fn.argtypes = res, args
                _libs[flavour] = L
    return L


def lib_for(precision):
    return lib(FLAVOUR.get(precision, ""))


def check(rc, what, L=None):
    if rc != 0:
        # the message lives in the library (flavour) that failed: the one named, else the first loaded one that has a message
        msg = ""
        for cand in ([L] if L is not None else list(_libs.values()) or [lib()]):
            msg = cand.ap_last_error().decode("utf-8", "replace")
            if msg:
                break
        raise RuntimeError("airpose_hip %s failed (status %d): %s" % (what, rc, msg))


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
