"""Drop-in for the reference's body-model object (boundary #2).

Mirrors ``smplx.SMPLX`` as the reference uses it (copenet/src/copenet/copenet_twoview.py:36-45 ctor,
:237-241 ``forward(betas=, body_pose=, global_orient=, transl=, pose2rot=False)``, :64-65 ``.to()``,
:69 ``.v_template``, :77 ``.faces``): the forward pass runs in libairpose_hip.so (pose prep + kinematic
chain, blend-shape contraction in split-bf16 form on the bf16 matrix pipe fused with the sparse skinning, joint/landmark gather).  Semantics follow
upstream smplx 0.1.28 (the fork's source is absent from the reference checkout).
"""
import ctypes
import threading

import numpy as np
import torch
import torch.nn as nn

from . import _native as N
from . import smplx_model as SM


class ModelOutput(object):
    """Attribute bag with the upstream output field names."""

    def __init__(self, **kw):
        self.vertices = self.joints = self.full_pose = self.betas = self.global_orient = None
        self.body_pose = self.expression = self.transl = None
        self.__dict__.update(kw)


class SMPLX(nn.Module):
    NUM_BODY_JOINTS = 21
    NUM_JOINTS = 55

    def __init__(self, model_path=None, batch_size=1, create_transl=False, gender="neutral", model_data=None,
                 num_betas=10, num_expression_coeffs=10, use_pca=True, num_pca_comps=6, flat_hand_mean=False, **kwargs):
        """model_path: directory holding SMPLX_{GENDER}.npz or the file itself (reference call sites);
        model_data: dict from smplx_model.make_synthetic_model / load_model_npz (tests, bench)."""
        super().__init__()
        if model_data is None:
            p = SM.find_model(model_path, gender)
            if p is None:
                raise FileNotFoundError("SMPL-X model file not found under %r (licence-gated download); pass "
                                        "model_data=smplx_model.make_synthetic_model() for a synthetic stand-in"
                                        % (model_path,))
            model_data = SM.load_model_npz(p, num_betas, num_expression_coeffs)
        self.batch_size = batch_size
        self.gender = gender
        self.num_betas, self.num_expression_coeffs = int(num_betas), int(num_expression_coeffs)
        # hand pose space of pose2rot=True calls (upstream SMPLH/SMPLX ctor defaults: 6 PCA components, non-flat mean)
        self.use_pca, self.num_pca_comps, self.flat_hand_mean = bool(use_pca), int(num_pca_comps), bool(flat_hand_mean)
        if not 1 <= self.num_pca_comps <= 45:
            raise ValueError("num_pca_comps must be in 1..45")
        self._md = {k: (np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v) for k, v in model_data.items()}
        self.faces = self._md["faces"]                                   # ndarray, as upstream
        self.register_buffer("faces_tensor", torch.from_numpy(self._md["faces"].astype(np.int64)))
        self.register_buffer("v_template", torch.from_numpy(self._md["v_template"]))
        self.num_verts = int(self._md["v_template"].shape[0])
        self._handle = None
        self._hdev = None
        self._lock = threading.Lock()

    def _native(self, device):
        N.require_gpu()
        if self._handle is not None and self._hdev == device.index:
            return self._handle
        if self._handle is not None:
            N.lib().ap_smplx_destroy(self._handle)
            self._handle = None
        md = self._md
        keep = dict(
            v_template=np.ascontiguousarray(md["v_template"], np.float32),
            shapedirs=np.ascontiguousarray(md["shapedirs"], np.float32),
            posedirs=np.ascontiguousarray(md["posedirs"], np.float32),
            J_regressor=np.ascontiguousarray(md["J_regressor"], np.float32),
            parents=np.ascontiguousarray(md["parents"], np.int64),
            lbs_weights=np.ascontiguousarray(md["lbs_weights"], np.float32),
            faces=np.ascontiguousarray(md["faces"], np.int64),
            extra_joint_verts=np.ascontiguousarray(md["extra_joint_verts"], np.int64),
            lmk_faces_idx=np.ascontiguousarray(md["lmk_faces_idx"], np.int64),
            lmk_bary_coords=np.ascontiguousarray(md["lmk_bary_coords"], np.float32))
        V, J = keep["v_template"].shape[0], keep["J_regressor"].shape[0]
        if keep["shapedirs"].shape != (V, 3, 20) or keep["posedirs"].shape != ((J - 1) * 9, V * 3):
            raise RuntimeError("SMPL-X model arrays have unexpected shapes: shapedirs %s posedirs %s"
                               % (keep["shapedirs"].shape, keep["posedirs"].shape))
        s = N.SmplxModelStruct()
        s.num_verts, s.num_joints, s.num_faces = V, J, keep["faces"].shape[0]
        s.num_shape_coeffs, s.num_extra, s.num_landmarks = 20, len(keep["extra_joint_verts"]), len(keep["lmk_faces_idx"])
        for k, a in keep.items():
            setattr(s, k, a.ctypes.data_as(ctypes.c_void_p))
        h = ctypes.c_void_p()
        N.check(N.lib().ap_smplx_create(ctypes.byref(h), ctypes.byref(s), device.index or 0), "ap_smplx_create")
        self._handle, self._hdev = h, device.index
        self.num_joints_out = N.lib().ap_smplx_num_joints_out(h)
        return h

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                N.lib().ap_smplx_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    @staticmethod
    def _rot(t, B, n, name, dev):
        if t is None:
            return None
        t = N.f32c(t, dev)
        if t.numel() != B * n * 9:
            raise RuntimeError("%s must hold %d rotation matrices per body (pose2rot=False)" % (name, n))
        return t.reshape(B, n, 3, 3)

    def forward(self, betas=None, global_orient=None, body_pose=None, left_hand_pose=None, right_hand_pose=None,
                transl=None, expression=None, jaw_pose=None, leye_pose=None, reye_pose=None, return_verts=True,
                return_full_pose=False, pose2rot=True, **kwargs):
        if pose2rot:
            # Axis-angle inputs (the upstream default; the reference's dataset code, aerialpeople.py:56-64).  smplx 0.1.28
            # SMPLX.forward: hands = pca_coeffs @ hands_components[:num_pca_comps] when use_pca, full_pose += pose_mean (the
            # model file's mean hand pose unless flat_hand_mean; zero for every other joint), un-supplied poses are the
            # module's zero parameters, then lbs.batch_rodrigues.  The hand arrays come with the model file
            # (hands_mean{l,r}, hands_components{l,r}); a model dict without them can only serve flat hands.
            from . import lbs
            ref = next((t for t in (body_pose, global_orient, betas) if t is not None), None)
            if ref is None or not ref.is_cuda:
                raise RuntimeError("airpose_amd.SMPLX: inputs must be CUDA (ROCm) tensors; there is no CPU path")
            dev0, Bp = ref.device, (body_pose.shape[0] if body_pose is not None else ref.shape[0])

            def aa(t, n, name):
                if t is None:
                    return None
                if t.dim() < 2 or t.shape[0] != Bp or t.numel() != Bp * n * 3 or not t.is_cuda:
                    raise RuntimeError("%s must be CUDA axis-angle vectors (B, %d), pose2rot=True" % (name, n * 3))
                return lbs.batch_rodrigues(t.reshape(-1, 3)).reshape(-1, n, 3, 3)

            def hand(t, side):
                mean, comp = self._md.get("hands_mean" + side), self._md.get("hands_components" + side)
                have = mean is not None and comp is not None
                if t is None and (self.flat_hand_mean or not have):
                    return None                               # identity hand rotations (flat_hand_mean=True semantics)
                if not have:
                    raise RuntimeError("hand poses with pose2rot=True need hands_mean%s / hands_components%s of the "
                                       "SMPL-X model file; this model dict carries none" % (side, side))
                width = self.num_pca_comps if self.use_pca else 45
                if t is None:
                    t = torch.zeros(Bp, width, device=dev0)
                if tuple(t.shape) != (Bp, width) or not t.is_cuda:
                    raise RuntimeError("%s_hand_pose must be CUDA (B, %d) (%s), got %s" % (
                        "left" if side == "l" else "right", width,
                        "PCA coefficients, use_pca=True" if self.use_pca else "axis-angle, use_pca=False", tuple(t.shape)))
                t = t.to(torch.float32)
                if self.use_pca:
                    t = t @ torch.from_numpy(np.ascontiguousarray(comp[:width])).to(dev0)
                if not self.flat_hand_mean:
                    t = t + torch.from_numpy(np.ascontiguousarray(mean)).to(dev0)
                return lbs.batch_rodrigues(t.reshape(-1, 3)).reshape(-1, 15, 3, 3)
            global_orient, body_pose = aa(global_orient, 1, "global_orient"), aa(body_pose, 21, "body_pose")
            jaw_pose, leye_pose, reye_pose = aa(jaw_pose, 1, "jaw_pose"), aa(leye_pose, 1, "leye_pose"), aa(reye_pose, 1, "reye_pose")
            left_hand_pose, right_hand_pose = hand(left_hand_pose, "l"), hand(right_hand_pose, "r")
        if betas is None or body_pose is None:
            raise RuntimeError("betas and body_pose are required (the reference creates no learnable defaults "
                               "on this path: create_transl=False, copenet_twoview.py:36-45)")
        if not betas.is_cuda:
            raise RuntimeError("airpose_amd.SMPLX: inputs must be CUDA (ROCm) tensors; there is no CPU path")
        dev = betas.device
        B = max(betas.shape[0], body_pose.shape[0])
        betas = N.f32c(betas, dev)
        if betas.shape[0] != B:
            betas = betas.expand(B, -1).contiguous()
        body = self._rot(body_pose, B, 21, "body_pose", dev)
        go = self._rot(global_orient, B, 1, "global_orient", dev)
        extra = None
        parts = (jaw_pose, leye_pose, reye_pose, left_hand_pose, right_hand_pose)
        if any(p is not None for p in parts):
            eye = torch.eye(3, device=dev).expand(B, 1, 3, 3)
            sizes = (1, 1, 1, 15, 15)
            extra = torch.cat([self._rot(p, B, n, "extra pose", dev) if p is not None else eye.expand(B, n, 3, 3)
                               for p, n in zip(parts, sizes)], dim=1).contiguous()
        expression = N.f32c(expression, dev)
        transl = N.f32c(transl, dev)
        if tuple(betas.shape) != (B, self.num_betas):
            raise RuntimeError("betas must be (1|B, %d), got %s" % (self.num_betas, tuple(betas.shape)))
        if expression is not None and tuple(expression.shape) != (B, self.num_expression_coeffs):
            raise RuntimeError("expression must be (B, %d), got %s" % (self.num_expression_coeffs, tuple(expression.shape)))
        if transl is not None and tuple(transl.shape) != (B, 3):
            raise RuntimeError("transl must be (B, 3), got %s" % (tuple(transl.shape),))
        verts = torch.empty(B, self.num_verts, 3, device=dev, dtype=torch.float32)
        with self._lock, torch.cuda.device(dev):
            h = self._native(dev)
            joints = torch.empty(B, self.num_joints_out, 3, device=dev, dtype=torch.float32)
            N.check(N.lib().ap_smplx_fwd(h, B, N.dptr(betas), N.dptr(expression), N.dptr(go), N.dptr(body),
                                         N.dptr(extra), N.dptr(transl), N.dptr(verts), N.dptr(joints),
                                         N.stream_ptr(dev)), "ap_smplx_fwd")
        return ModelOutput(vertices=verts if return_verts else None, joints=joints, betas=betas, expression=expression,
                           global_orient=global_orient, body_pose=body_pose, transl=transl)

    def forward_fused(self, pred_pose, pred_betas, cam_center=None, focal_length=(1475.0, 1475.0), want_rotmat=True):
        """rot6d -> SMPL-X (global_orient = I, transl = 0) -> transform_smpl([R_root | trans]) -> projection for
        one view in three launches + one GEMM (copenet_twoview.py:222-223, 237-246, 307-311).
        pred_pose (n,135) with the translation already un-scaled.  Returns dict."""
        if not pred_pose.is_cuda:
            raise RuntimeError("airpose_amd.SMPLX: inputs must be CUDA (ROCm) tensors; there is no CPU path")
        dev = pred_pose.device
        n = pred_pose.shape[0]
        pred_pose, pred_betas = N.f32c(pred_pose), N.f32c(pred_betas, dev)
        cam_center = N.f32c(cam_center, dev)
        verts = torch.empty(n, self.num_verts, 3, device=dev, dtype=torch.float32)
        with self._lock, torch.cuda.device(dev):
            h = self._native(dev)
            nj = self.num_joints_out
            joints = torch.empty(n, nj, 3, device=dev, dtype=torch.float32)
            j2d = torch.empty(n, nj, 2, device=dev, dtype=torch.float32) if cam_center is not None else None
            rot = torch.empty(n, 22, 3, 3, device=dev, dtype=torch.float32) if want_rotmat else None
            N.check(N.lib().ap_smplx_fwd_fused(h, n, N.dptr(pred_pose), pred_pose.shape[1], N.dptr(pred_betas),
                                               N.dptr(cam_center), float(focal_length[0]), float(focal_length[1]),
                                               N.dptr(verts), N.dptr(joints), N.dptr(j2d), N.dptr(rot),
                                               N.stream_ptr(dev)), "ap_smplx_fwd_fused")
        return {"vertices_cam": verts, "j3d_cam": joints, "j2d_cam": j2d, "rotmat": rot}

    def forward_twoview(self, pred_pose, pred_betas, intr0, intr1, trans_scale=0.0, in_smpltrans=None,
                        focal_length=(1475.0, 1475.0), want_rotmat=True):
        """The caller slice for both views in ONE native call (ap_smplx_fwd_twoview; copenet_twoview.py:214-223,
        237-279, 307-317).  pred_pose (2,B,135) / pred_betas (2,B,10): view 0 first, contiguous; with trans_scale > 0
        the translation columns of pred_pose are un-scaled IN PLACE (the reference's ``pred_smpltrans /= trans_scale``
        on a view of pred_pose).  in_smpltrans (2,B,3): also emit the test-mode input meshes (betas = 0, identity
        root).  Returns dict of (2,B,...) tensors."""
        if not pred_pose.is_cuda:
            raise RuntimeError("airpose_amd.SMPLX: inputs must be CUDA (ROCm) tensors; there is no CPU path")
        dev = pred_pose.device
        if pred_pose.dim() != 3 or pred_pose.shape[0] != 2 or pred_pose.shape[2] != 135 or not pred_pose.is_contiguous() \
                or pred_pose.dtype != torch.float32:
            raise RuntimeError("forward_twoview: pred_pose must be a contiguous fp32 (2, B, 135) tensor")
        B = pred_pose.shape[1]
        pred_betas = N.f32c(pred_betas, dev)
        if tuple(pred_betas.shape) != (2, B, 10):
            raise RuntimeError("forward_twoview: pred_betas must be (2, B, 10)")
        intr0, intr1 = N.f32c(intr0, dev), N.f32c(intr1, dev)
        for t in (intr0, intr1):
            if t is not None and tuple(t.shape) != (B, 3, 3):
                raise RuntimeError("forward_twoview: intr must be (B, 3, 3)")
        in_smpltrans = N.f32c(in_smpltrans, dev)
        if in_smpltrans is not None and tuple(in_smpltrans.shape) != (2, B, 3):
            raise RuntimeError("forward_twoview: in_smpltrans must be (2, B, 3)")
        nv = 4 if in_smpltrans is not None else 2
        verts = torch.empty(nv, B, self.num_verts, 3, device=dev, dtype=torch.float32)
        with self._lock, torch.cuda.device(dev):
            h = self._native(dev)
            nj = self.num_joints_out
            joints = torch.empty(2, B, nj, 3, device=dev, dtype=torch.float32)
            j2d = torch.empty(2, B, nj, 2, device=dev, dtype=torch.float32) if intr0 is not None else None
            rot = torch.empty(2, B, 22, 3, 3, device=dev, dtype=torch.float32) if want_rotmat else None
            N.check(N.lib().ap_smplx_fwd_twoview(h, B, N.dptr(pred_pose), 135, float(trans_scale), N.dptr(pred_betas),
                                                 N.dptr(intr0), N.dptr(intr1), float(focal_length[0]),
                                                 float(focal_length[1]), N.dptr(in_smpltrans), N.dptr(verts),
                                                 N.dptr(joints), N.dptr(j2d), N.dptr(rot), N.stream_ptr(dev)),
                    "ap_smplx_fwd_twoview")
        return {"vertices_cam": verts[:2], "vertices_cam_in": verts[2:] if nv == 4 else None, "j3d_cam": joints,
                "j2d_cam": j2d, "rotmat": rot}

    def set_blend_precision(self, precision):
        """'bf16x2' (default): blend-shape contraction as split-bf16 products on the bf16 matrix pipe; 'fp32': exact
        fp32 MFMA chain (4x slower; the two differ by ~1e-7 of the vertex scale)."""
        N.check(N.lib().ap_smplx_set_blend_precision(self._native(torch.device("cuda", torch.cuda.current_device())),
                                                     N.PRECISIONS[precision]), "ap_smplx_set_blend_precision")

    def set_fused(self, on):
        """Blend-shape contraction + skinning in one kernel where the call allows it (1, default), always as two kernels (0);
        4: the fused kernel with the joints stage inside it (A/B aid; slower)."""
        N.check(N.lib().ap_smplx_set_fused(self._native(torch.device("cuda", torch.cuda.current_device())), int(on)),
                "ap_smplx_set_fused")

    def enable_timing(self, on=True):
        """True / 1: per-stage events; 2: only the span of the whole tail (timing()["prep_ms"] then holds it); False: off."""
        N.check(N.lib().ap_smplx_enable_timing(self._native(torch.device("cuda", torch.cuda.current_device())),
                                               int(on)), "ap_smplx_enable_timing")

    def timing(self, reset=True):
        ms = (ctypes.c_double * 4)()
        n = ctypes.c_int64()
        N.check(N.lib().ap_smplx_timing(self._handle, ms, ctypes.byref(n), int(reset)), "ap_smplx_timing")
        return {"prep_ms": ms[0], "blend_gemm_ms": ms[1], "skin_ms": ms[2], "joints_ms": ms[3], "passes": n.value}
