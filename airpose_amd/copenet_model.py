"""Drop-in for the reference network module (boundary #1).

Mirrors ``copenet.models.model_copenet`` (copenet/src/copenet/models/model_copenet.py):
``copenet(block, layers, smpl_mean_params)`` with ``forward`` (:112-159), ``forward_feat_ext``
(:161-176), ``forward_reg`` (:178-204) and ``getcopenet`` (:229-239) -- same argument names,
same outputs, same sub-module / buffer names, hence the same 331 ``state_dict`` keys, so
``copenet_twoview`` (copenet/src/copenet/copenet_twoview.py:60,79-80) and Lightning checkpoints
work unchanged.  The nn.Conv2d / nn.BatchNorm2d / nn.Linear children are parameter containers
only: all compute goes through the C ABI of libairpose_hip.so (hand-written gfx950 kernels).
Inference (eval) only; there is no CPU or eager fallback.
"""
import ctypes
import threading

import numpy as np
import torch
import torch.nn as nn

from . import _native as N


class Bottleneck(nn.Module):
    """Parameter container with the reference block's child names (model_copenet.py:8-25)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError("airpose_amd.Bottleneck holds parameters only; the trunk runs in libairpose_hip.so")


class copenet(nn.Module):
    """SMPL-X iterative regressor with ResNet-50 trunk and cross-view fusion, on MI355X."""

    variant = 0            # ap_net_create variant (0: two-view copenet)
    AUTO_ORDER = ("f16", "bf16", "bf16x2", "fp32")          # fastest first (bench.py: 47k / 48k / 15.7k / 6.9k pairs/s at 256 pairs)
    AUTO_BAR = 1e-4                                          # north_star: outputs within 1e-4 of the fp32 path
    AUTO_MARGIN = 0.75                                       # a mode is kept when its probe is below AUTO_MARGIN x AUTO_BAR: over 36 (mode, checkpoint)
                                                             # cells the whole-pipeline error against the fp32 CPU oracle was 0.76 .. 1.34 x the probe's
                                                             # (profiles/r06_probe_vs_oracle.txt, tools/probe_vs_oracle.py)
    AUTO_PAIRS = 8
    fc1_extra = 3 + 3 + 6 + 21 * 6 + 10 + 21 * 6 + 10

    def __init__(self, block=Bottleneck, layers=(3, 4, 6, 3), smpl_mean_params=None, precision="f16"):
        super().__init__()
        if tuple(layers) != (3, 4, 6, 3):
            raise ValueError("only the ResNet-50 layout [3, 4, 6, 3] of the reference is supported")
        if precision not in N.PRECISIONS and precision != "auto":
            raise ValueError("precision must be 'bf16' / 'f16' (throughput; bf16 or fp16 storage), 'bf16x2' (split-bf16: fast parity mode), "
                             "'fp32' (exact fp32 MFMA chain) or 'auto' (the fastest of these whose parity probe holds AUTO_BAR on the "
                             "loaded checkpoint)")
        # precision="auto": resolved when the weights are packed (first forward after load_state_dict), by parity_probe() of each
        # candidate in AUTO_ORDER on THIS checkpoint; self.precision is then the chosen mode, self.auto_report what every tried one measured
        if precision == "auto" and self.variant != 0:
            raise ValueError("precision='auto' needs the two-view copenet head (ap_net_parity_probe runs copenet.forward); "
                             "pick a mode for the hmr / muhmr / copenet_singleview heads")
        self.precision_requested = precision
        self.precision = self.AUTO_ORDER[0] if precision == "auto" else precision
        self.auto_report = None
        self.inplanes = 64
        npose = 21 * 6
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc1 = nn.Linear(512 * 4 + self.fc1_extra, 1024)
        self.drop1 = nn.Dropout()
        self.fc2 = nn.Linear(1024, 1024)
        self.drop2 = nn.Dropout()
        self.decpose = nn.Linear(1024, self._npose_out(npose))
        self.decshape = nn.Linear(1024, 10)
        self.deccam = nn.Linear(1024, 3)
        for m in (self.decpose, self.decshape, self.deccam):
            nn.init.xavier_uniform_(m.weight, gain=0.01)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, (2.0 / n) ** 0.5)
        if smpl_mean_params is None:
            pose, shape, cam = np.zeros(144, np.float32), np.zeros(10, np.float32), np.zeros(3, np.float32)
        else:
            mp = np.load(smpl_mean_params)
            pose, shape, cam = mp["pose"][:], mp["shape"][:].astype("float32"), mp["cam"]
        self.register_buffer("init_pose", torch.from_numpy(np.asarray(pose, np.float32)).unsqueeze(0))
        self.register_buffer("init_shape", torch.from_numpy(np.asarray(shape, np.float32)).unsqueeze(0))
        self.register_buffer("init_cam", torch.from_numpy(np.asarray(cam, np.float32)).unsqueeze(0))
        self._handle = None
        self._packed_sig = None
        self._lock = threading.Lock()   # handles are not re-entrant (rospy callbacks come from other threads)

    @staticmethod
    def _npose_out(npose):
        return 3 + 6 + npose

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        mods = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        mods += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    # ------------------------------------------------------------------ native handle
    def _signature(self):
        """Cheap change detector for the packed copy: per tensor the autograd version counter, the storage address
        and the object identity.  It sees in-place ops, load_state_dict, .to() and re-assigned parameters; it CANNOT
        see a write through ``.data`` / ``.detach()`` into the same storage (``p.data.copy_(w)``, the idiom of the
        reference's own init, model_copenet.py:74-84) -- call ``repack()`` after such writes."""
        sig = [self.precision]
        for t in list(self.parameters()) + list(self.buffers()):
            sig.append((t._version, t.data_ptr(), id(t)))
        return hash(tuple(sig))

    def repack(self):
        """Force the next forward to re-pack every tensor into the native handle (after writes through ``.data``)."""
        self._packed_sig = None

    def _native(self, device):
        """(Re)pack the current parameters into the library when they changed.  precision="auto": pack as each candidate mode in turn
        (fastest first) and keep the first whose GPU parity probe against the exact-fp32 trunk of the same weights holds AUTO_BAR."""
        N.require_gpu()
        sig = (device.index, self._signature())
        if self._handle is not None and sig == self._packed_sig:
            return self._handle
        if self.precision_requested != "auto":
            return self._pack(device, sig)
        report = {}
        for prec in self.AUTO_ORDER:
            if self._handle is not None and prec != self.precision:
                self._L().ap_net_destroy(self._handle)       # another storage type: another handle
                self._handle = None
            self.precision = prec
            sig = (device.index, self._signature())
            try:
                self._pack(device, sig)
                report[prec] = self._probe_locked(device, self.AUTO_PAIRS, 20240601)
            except N.RangeError as e:                        # fp16 storage cannot hold this checkpoint (weights or probe activations)
                report[prec] = {"error": str(e)[:200], "holds": False}
                if self._handle is not None:
                    self._L().ap_net_range_status(self._handle, N.stream_ptr(device), 1)
                continue
            except RuntimeError as e:
                if "exceeds the fp16 range" not in str(e):
                    raise
                report[prec] = {"error": str(e)[:200], "holds": False}
                continue
            report[prec]["holds"] = report[prec]["max_rel_err"] < self.AUTO_MARGIN * self.AUTO_BAR
            if report[prec]["holds"]:
                break
        self.auto_report = {"chosen": self.precision, "bar": self.AUTO_BAR, "probe_must_be_below": self.AUTO_MARGIN * self.AUTO_BAR,
                            "pairs": self.AUTO_PAIRS, "tried": report}
        return self._handle

    def _pack(self, device, sig):
        L = self._L()
        if self._handle is not None and getattr(self, "_hdev", device.index) != device.index:
            # the handle (packed weights, workspace, per-device launch state) belongs to the device of the first call
            L.ap_net_destroy(self._handle)
            self._handle = None
        if self._handle is None:
            h = ctypes.c_void_p()
            N.check(L.ap_net_create(ctypes.byref(h), device.index or 0, N.PRECISIONS[self.precision], self.variant),
                    "ap_net_create")
            self._handle, self._hdev = h, device.index
            for entry, value in getattr(self, "_knobs", {}).items():     # knobs set on the previous handle
                N.check(getattr(L, entry)(self._handle, value), entry)
        for name, t in self.state_dict().items():
            if not t.dtype.is_floating_point:
                continue                                   # num_batches_tracked
            a = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
            shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
            N.check(L.ap_net_set_tensor(self._handle, name.encode(), a.ctypes.data_as(ctypes.c_void_p), shape, a.ndim),
                    "ap_net_set_tensor(%s)" % name)
        N.check(L.ap_net_finalize(self._handle), "ap_net_finalize")
        self._packed_sig = sig
        return self._handle

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                self._L().ap_net_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def _check_eval(self):
        if self.training:
            raise RuntimeError("airpose_amd.copenet implements the inference path only: call .eval() "
                               "(training through the HIP kernels is out of scope)")

    @staticmethod
    def _dev(x):
        if not x.is_cuda:
            raise RuntimeError("airpose_amd.copenet: inputs must be CUDA (ROCm) tensors; there is no CPU path")
        return x.device

    # ------------------------------------------------------------------ reference API
    def forward_feat_ext(self, x):
        """(n,3,224,224) -> (n,2048)   [model_copenet.py:161-176]"""
        self._check_eval()
        dev = self._dev(x)
        if x.dim() != 4 or x.shape[1:] != (3, 224, 224):
            raise RuntimeError("forward_feat_ext expects (n, 3, 224, 224) NCHW crops (AvgPool2d(7) fixes the size)")
        x = N.f32c(x)
        out = torch.empty(x.shape[0], 2048, device=dev, dtype=torch.float32)
        with self._lock, torch.cuda.device(dev):
            h = self._native(dev)
            N.check(self._L().ap_trunk_fwd(h, N.dptr(x, "x"), x.shape[0], N.dptr(out), N.stream_ptr(dev)), "ap_trunk_fwd")
        return out

    def forward_feat_ext_twoview(self, x0, x1, out=None, out_stream=None):
        """xf0, xf1 of forward() (model_copenet.py:140-141) in one native call: two (B,3,224,224) -> (2,B,2048); the two views run
        as two concurrent trunk passes exactly as inside forward().  ``out``: a caller-owned (2,B,2048) fp32 buffer.
        ``out_stream``: a torch stream on which the features become ready INSTEAD of the current one (the inputs are still taken
        in the order of the current stream, which is then not made to wait for the trunk: ap_trunk_fwd_twoview_async)."""
        self._check_eval()
        dev = self._dev(x0)
        B = x0.shape[0]
        if x0.dim() != 4 or x0.shape[1:] != (3, 224, 224) or x1.shape != x0.shape:
            raise RuntimeError("forward_feat_ext_twoview expects two (B, 3, 224, 224) NCHW crops")
        x0_in, x1_in = x0, x1
        x0, x1 = N.f32c(x0), N.f32c(x1, dev)
        if out is None:
            out = torch.empty(2, B, 2048, device=dev, dtype=torch.float32)
        elif tuple(out.shape) != (2, B, 2048) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
            raise RuntimeError("out must be a contiguous (2, %d, 2048) fp32 tensor on %s" % (B, dev))
        with self._lock, torch.cuda.device(dev):
            h = self._native(dev)
            if out_stream is None:
                N.check(self._L().ap_trunk_fwd_twoview(h, N.dptr(x0, "x0"), N.dptr(x1, "x1"), B, N.dptr(out), N.stream_ptr(dev)),
                        "ap_trunk_fwd_twoview")
            else:
                N.check(self._L().ap_trunk_fwd_twoview_async(h, N.dptr(x0, "x0"), N.dptr(x1, "x1"), B, N.dptr(out), N.stream_ptr(dev),
                                                             ctypes.c_void_p(out_stream.cuda_stream)), "ap_trunk_fwd_twoview_async")
                # The current stream is NOT behind the passes: an fp32 / contiguous COPY made above (half, uint8, strided crops) would
                # go back to the caching allocator when this function returns and could be handed out again on the current stream
                # while the passes still read it.  out_stream is behind the passes (the joins are queued): tie the copies to it.
                for conv, orig in ((x0, x0_in), (x1, x1_in)):
                    if conv is not orig:
                        conv.record_stream(out_stream)
        return out

    @staticmethod
    def _bs(t, B, width, name):
        """(tensor, batch stride) for an optional (1|B, >=width) initial-state tensor."""
        if t is None:
            return None, 0
        if t.dim() != 2 or t.shape[1] < width or t.shape[0] not in (1, B):
            raise RuntimeError("%s must be (1|B, >=%d)" % (name, width))
        return t, (0 if t.shape[0] == 1 and B != 1 else t.shape[1])

    def _ief(self, entry, a0, a1, bb0, bb1, pos0, pos1, th0, th1, sh0, sh1, B, iters, dev):
        bb0, bb1, pos0, pos1 = (N.f32c(t, dev) for t in (bb0, bb1, pos0, pos1))
        for name, t in (("bb0", bb0), ("bb1", bb1), ("init_position0", pos0), ("init_position1", pos1)):
            if tuple(t.shape) != (B, 3):                     # torch.cat in forward_reg would raise on these
                raise RuntimeError("%s must be (%d, 3), got %s" % (name, B, tuple(t.shape)))
        if a0.shape[0] != B or a1.shape != a0.shape:
            raise RuntimeError("the two views must hold the same number of samples")
        if int(iters) < 1:
            raise RuntimeError("iters must be >= 1 (forward always evaluates the regressor once)")
        th0, th0s = self._bs(N.f32c(th0, dev), B, 132, "init_theta0")
        th1, th1s = self._bs(N.f32c(th1, dev), B, 132, "init_theta1")
        sh0, sh0s = self._bs(N.f32c(sh0, dev), B, 10, "init_shape0")
        sh1, sh1s = self._bs(N.f32c(sh1, dev), B, 10, "init_shape1")
        pose = torch.empty(2, B, 135, device=dev, dtype=torch.float32)
        betas = torch.empty(2, B, 10, device=dev, dtype=torch.float32)
        with self._lock, torch.cuda.device(dev):
            h = self._native(dev)
            fn = getattr(self._L(), entry)
            N.check(fn(h, N.dptr(a0), N.dptr(a1), N.dptr(bb0), N.dptr(bb1), N.dptr(pos0), N.dptr(pos1),
                       N.dptr(th0), th0s, N.dptr(th1), th1s, N.dptr(sh0), sh0s, N.dptr(sh1), sh1s, B, int(iters),
                       N.dptr(pose[0]), N.dptr(betas[0]), N.dptr(pose[1]), N.dptr(betas[1]), N.stream_ptr(dev)), entry)
        return pose[0], betas[0], pose[1], betas[1]

    def forward(self, x0, x1, bb0, bb1, init_position0, init_position1, init_theta0=None, init_theta1=None,
                init_shape0=None, init_shape1=None, iters=3):
        """[model_copenet.py:112-159] -> (pred_pose0 (B,135), pred_betas0 (B,10), pred_pose1, pred_betas1)."""
        self._check_eval()
        dev = self._dev(x0)
        B = x0.shape[0]
        if x0.shape[1:] != (3, 224, 224) or x1.shape != x0.shape:
            raise RuntimeError("forward expects two (B, 3, 224, 224) crops")
        return self._ief("ap_copenet_fwd", N.f32c(x0), N.f32c(x1, dev), bb0, bb1, init_position0, init_position1,
                         init_theta0, init_theta1, init_shape0, init_shape1, B, iters, dev)

    def forward_ief(self, xf0, xf1, bb0, bb1, init_position0, init_position1, init_theta0=None, init_theta1=None,
                    init_shape0=None, init_shape1=None, iters=3):
        """The IEF loop of forward() from pre-computed trunk features (model_copenet.py:144-157)."""
        self._check_eval()
        dev = self._dev(xf0)
        if xf0.dim() != 2 or xf0.shape[1] != 2048 or xf1.shape != xf0.shape:
            raise RuntimeError("forward_ief expects two (B, 2048) feature tensors")
        return self._ief("ap_regressor_fwd", N.f32c(xf0), N.f32c(xf1, dev), bb0, bb1, init_position0, init_position1,
                         init_theta0, init_theta1, init_shape0, init_shape1, xf0.shape[0], iters, dev)

    def forward_reg(self, xf0, xf1, bb0, bb1, pred_position0, pred_position1, pred_orient0, pred_orient1,
                    pred_art_pose0, pred_art_pose1, pred_shape0, pred_shape1):
        """One regressor evaluation for both views [model_copenet.py:178-204]."""
        th0 = torch.cat([pred_orient0, pred_art_pose0], 1)
        th1 = torch.cat([pred_orient1, pred_art_pose1], 1)
        return self.forward_ief(xf0, xf1, bb0, bb1, pred_position0, pred_position1, th0, th1,
                                pred_shape0, pred_shape1, iters=1)

    def regressor_step(self, xf, bb, pose, betas, partner):
        """One forward_reg evaluation for ONE view with the partner's (art_pose | shape) (B,136) supplied by
        the caller -- the view-split / on-drone exchange step (README.md:238-241)."""
        self._check_eval()
        dev = self._dev(xf)
        B = xf.shape[0]
        xf, bb, pose, betas, partner = (N.f32c(t, dev) for t in (xf, bb, pose, betas, partner))
        if partner.shape != (B, 136) or pose.shape != (B, 135) or betas.shape != (B, 10):
            raise RuntimeError("regressor_step: pose (B,135), betas (B,10), partner (B,136)")
        if xf.shape != (B, 2048) or bb.shape != (B, 3):
            raise RuntimeError("regressor_step: xf (B,2048), bb (B,3)")
        pose_out = torch.empty(B, 135, device=dev, dtype=torch.float32)
        betas_out = torch.empty(B, 10, device=dev, dtype=torch.float32)
        with self._lock, torch.cuda.device(dev):
            h = self._native(dev)
            N.check(self._L().ap_regressor_step(h, N.dptr(xf), N.dptr(bb), N.dptr(pose), N.dptr(betas), N.dptr(partner),
                                              136, B, N.dptr(pose_out), N.dptr(betas_out), N.stream_ptr(dev)),
                    "ap_regressor_step")
        return pose_out, betas_out

    PROBE_SLICES = ("theta.trans", "theta.rot6d", "betas", "proj(trans)")

    def _probe_locked(self, dev, n_pairs, seed):
        err = (ctypes.c_double * 8)()
        import time
        t0 = time.perf_counter()
        N.check(self._L().ap_net_parity_probe(self._handle, int(n_pairs), ctypes.c_uint64(seed), err, N.stream_ptr(dev)),
                "ap_net_parity_probe")
        ms = (time.perf_counter() - t0) * 1e3
        rel = dict(zip(self.PROBE_SLICES, err[0:4]))
        return {"max_rel_err": max(rel.values()), "rel_err_by_slice": rel, "elementwise_err_by_slice": dict(zip(self.PROBE_SLICES, err[4:8])),
                "pairs": int(n_pairs), "seed": int(seed), "ms": ms, "precision": self.precision,
                "reference": "exact-fp32 trunk of the same weights on the GPU (ap_net_parity_probe)"}

    def parity_probe(self, n_pairs=8, seed=20240601):
        """What this handle's precision costs on the LOADED checkpoint: a seeded probe batch through the handle's trunk and through an
        exact-fp32 trunk packed from the same weights, both followed by the fp32 regressor (3 IEF iterations), compared on the host
        in fp64.  Returns the norm-wise error per slice (translation, 6-D rotations, betas, projected translation), the element-wise
        error and the wall time; the first call after a (re)load also packs the fp32 reference trunk."""
        self._check_eval()
        dev = torch.device("cuda", torch.cuda.current_device())
        with self._lock, torch.cuda.device(dev):
            self._native(dev)
            return self._probe_locked(dev, n_pairs, seed)

    def regressor_feat_part(self, xf):
        """bf + Wf[:, :2048] xf: the part of a folded regressor step that is constant over the IEF iterations -> (B, 148)."""
        self._check_eval()
        dev = self._dev(xf)
        xf = N.f32c(xf, dev)
        if xf.dim() != 2 or xf.shape[1] != 2048:
            raise RuntimeError("regressor_feat_part: xf (B,2048)")
        out = torch.empty(xf.shape[0], 148, device=dev, dtype=torch.float32)
        with self._lock, torch.cuda.device(dev):
            h = self._native(dev)
            N.check(self._L().ap_regressor_feat_part(h, N.dptr(xf), xf.shape[0], N.dptr(out), N.stream_ptr(dev)), "ap_regressor_feat_part")
        return out

    def regressor_step_local(self, hfeat, bb, pose, betas):
        """The partner-independent columns of one forward_reg step (model_copenet.py:185: xf | bb | pos | orient | art | shape): runs
        while the partner's 136 floats are still on the wire -> partial (B, 148)."""
        dev = self._dev(hfeat)
        B = hfeat.shape[0]
        hfeat, bb, pose, betas = (N.f32c(t, dev) for t in (hfeat, bb, pose, betas))
        if hfeat.shape != (B, 148) or bb.shape != (B, 3) or pose.shape != (B, 135) or betas.shape != (B, 10):
            raise RuntimeError("regressor_step_local: hfeat (B,148), bb (B,3), pose (B,135), betas (B,10)")
        out = torch.empty(B, 148, device=dev, dtype=torch.float32)
        with self._lock, torch.cuda.device(dev):
            h = self._native(dev)
            N.check(self._L().ap_regressor_step_local(h, N.dptr(hfeat), N.dptr(bb), N.dptr(pose), N.dptr(betas), B, N.dptr(out),
                                                      N.stream_ptr(dev)), "ap_regressor_step_local")
        return out

    def regressor_step_finish(self, partial, pose, betas, partner):
        """partial + the partner's 136 columns + the residual add -> (pose (B,135), betas (B,10)) of the step."""
        dev = self._dev(partial)
        B = partial.shape[0]
        partial, pose, betas, partner = (N.f32c(t, dev) for t in (partial, pose, betas, partner))
        if partial.shape != (B, 148) or pose.shape != (B, 135) or betas.shape != (B, 10) or partner.shape != (B, 136):
            raise RuntimeError("regressor_step_finish: partial (B,148), pose (B,135), betas (B,10), partner (B,136)")
        pose_out = torch.empty(B, 135, device=dev, dtype=torch.float32)
        betas_out = torch.empty(B, 10, device=dev, dtype=torch.float32)
        with self._lock, torch.cuda.device(dev):
            h = self._native(dev)
            N.check(self._L().ap_regressor_step_finish(h, N.dptr(partial), N.dptr(pose), N.dptr(betas), N.dptr(partner), 136, B,
                                                       N.dptr(pose_out), N.dptr(betas_out), N.stream_ptr(dev)), "ap_regressor_step_finish")
        return pose_out, betas_out

    # ------------------------------------------------------------------ fp16 range sentinel
    def range_mark_next(self, slot):
        """The next trunk-running call snapshots the range words of its pass streams into per-handle slot ``slot`` (behind its own
        last kernels: later batches cannot leak into it)."""
        dev = torch.device("cuda", torch.cuda.current_device())
        with self._lock:
            N.check(self._L().ap_net_range_mark_next(self._native(dev), int(slot)), "ap_net_range_mark_next")

    def range_slot(self, slot):
        """Raise RangeError if the snapshot of ``slot`` saw a range word set.  No synchronisation: wait for an event recorded behind
        the marked call first."""
        N.check(self._L().ap_net_range_slot(self._handle, int(slot)), "ap_net_range_slot")

    def range_peek(self):
        """Raise RangeError if the flag is set right now.  No synchronisation."""
        N.check(self._L().ap_net_range_peek(self._handle), "ap_net_range_peek")

    # ------------------------------------------------------------------ measurement hooks (bench.py)
    def enable_timing(self, on=True):
        N.check(self._L().ap_net_enable_timing(self._native(torch.device("cuda", torch.cuda.current_device())), int(on)),
                "ap_net_enable_timing")

    def timing(self, reset=True):
        ms = (ctypes.c_double * 4)()
        n = ctypes.c_int64()
        N.check(self._L().ap_net_timing(self._handle, ms, ctypes.byref(n), int(reset)), "ap_net_timing")
        return {"stem_ms": ms[0], "conv_ms": ms[1], "avgpool_ms": ms[2], "regressor_ms": ms[3], "passes": n.value}

    def last_conv_launches(self):
        """Kernel launches of the conv stack in the most recent trunk-running call (all passes): what the library chose."""
        return int(self._L().ap_net_last_conv_launches(self._handle))

    @staticmethod
    def _L():
        return N.lib()

    def set_range_check(self, mode):
        """precision="f16": 1 (default) deferred -- a non-finite trunk feature makes the NEXT call raise RangeError; 2 -- every
        forward synchronises its stream and raises for its own pass; 0 off."""
        self._set_knob("ap_net_set_range_check", mode)

    def range_status(self, reset=False):
        """Synchronise the current stream and raise RangeError if a trunk pass of this handle produced non-finite features
        (precision="f16": a stored activation left the fp16 range).  No-op for the other precisions."""
        dev = torch.device("cuda", torch.cuda.current_device())
        with self._lock, torch.cuda.device(dev):
            h = self._native(dev)
            N.check(self._L().ap_net_range_status(h, N.stream_ptr(dev), int(bool(reset))), "ap_net_range_status")

    def _set_knob(self, entry, value):
        """Per-handle knob: remembered on the module and re-applied whenever the native handle is re-created (a call
        with inputs on another GPU destroys the handle of the first device; its knobs must not fall back to defaults)."""
        self._knobs = dict(getattr(self, "_knobs", {}))
        self._knobs[entry] = int(value)
        with self._lock:
            h = self._native(torch.device("cuda", torch.cuda.current_device()))
            N.check(getattr(self._L(), entry)(h, int(value)), entry)

    def set_fold(self, on):
        """Evaluate fc1 -> fc2 -> dec as one folded affine map (default) or as the literal chain."""
        self._set_knob("ap_net_set_fold", on)

    def fold_status(self):
        """(state, probe_rel_err): state 1 = folded regressor map in use, 2 = literal chain by choice, 0 = literal chain because
        ap_net_finalize found the fold of THIS checkpoint off by more than 1e-5 on its probe batch."""
        err = ctypes.c_double()
        with self._lock:
            h = self._native(torch.device("cuda", torch.cuda.current_device()))
            st = self._L().ap_net_fold_status(h, ctypes.byref(err))
        return st, err.value

    def set_fuse_ief(self, on):
        """Folded map: all IEF iterations in one kernel (default) or one GEMM per iteration."""
        self._set_knob("ap_net_set_fuse_ief", on)

    def set_fuse_ds(self, on):
        self._set_knob("ap_net_set_fuse_ds", on)

    def set_fuse_block(self, on):
        """bf16 / f16: each layer1 bottleneck as one fused kernel (default, bottleneck2.hip) or as separate convolutions."""
        self._set_knob("ap_net_set_fuse_block", on)

    def set_fuse_pair(self, on):
        """bf16 / f16: conv3 of an identity block and conv1 of the next block as one pixel-local kernel (default) or two."""
        self._set_knob("ap_net_set_fuse_pair", on)

    def set_fuse_tail(self, on):
        """bf16 / f16: conv1 of layer2.0 inside the kernel of layer1's last bottleneck (default) or as its own convolution."""
        self._set_knob("ap_net_set_fuse_tail", on)

    def set_img_block(self, on):
        """bf16 / f16: layer3's identity bottlenecks as one image-resident kernel each: 1 (default) when the pass fills whole rounds
        of the chip, 2 always, 0 never (conv2 + fused pairs); same bits."""
        self._set_knob("ap_net_set_img_block", on)

    def set_img3(self, on):
        """bf16 / f16: conv2 of the layer2 identity blocks with half an image resident in LDS (conv_img3.hip): 1 (default) when the pass
        fills whole rounds of the chip, 2 always, 0 never (slab kernel); same bits."""
        self._set_knob("ap_net_set_img3", on)

    def set_pw_conv(self, on):
        """bf16 / f16: the pointwise layers of layer3 / layer4 no fused kernel covers on the one-wave-per-SIMD kernel (conv_pw.hip): 1 (default)
        when their tiles fill whole rounds of the chip (conv1 layers, conv3 + folded downsample; alone on the chip also the stride-2
        3 x 3 of a stage's first block); 2 whenever supported, conv3 + identity too; 3 conv1 whenever supported; 4 as 1 without the
        stride-2 3 x 3; 0 never (generic kernels).  Same bits in every setting."""
        self._set_knob("ap_net_set_pw_conv", on)

    def set_even_out(self, on):
        """bf16 / f16: block outputs only a stride-2 downsample reads are stored at the even pixels only (default) or in full."""
        self._set_knob("ap_net_set_even_out", on)

    def set_fuse_pool(self, on):
        """bf16 / f16: AvgPool2d(7) in the epilogue of the last convolution (default) or as its own kernel."""
        self._set_knob("ap_net_set_fuse_pool", on)

    def set_tiled(self, on):
        """bf16 / f16: pair-kernel-only intermediates in the pair kernel's fragment order (default) or NHWC; same bits."""
        self._set_knob("ap_net_set_tiled", on)

    def set_s2p(self, on):
        """conv2 of layer2.0 on the polyphase kernel conv_s2p.hip (1) / the generic stride-2 kernels (0, default)."""
        self._set_knob("ap_net_set_s2p", on)

    def set_fuse_stem(self, on):
        self._set_knob("ap_net_set_fuse_stem", on)

    def set_dual_stream(self, on):
        """Two-view forwards of >= 64 pairs: the two views as two concurrent trunk passes (default) or one pass."""
        self._set_knob("ap_net_set_dual_stream", on)

    def set_chunk(self, images):
        self._set_knob("ap_net_set_chunk", images)


def getcopenet(smpl_mean_params, pretrained=True, precision="f16", **kwargs):
    """model_copenet.getcopenet (:229-239).  ImageNet initialisation needs torchvision + network, neither
    of which exists here; weights arrive through load_state_dict / load_from_checkpoint instead."""
    return copenet(Bottleneck, [3, 4, 6, 3], smpl_mean_params, precision=precision, **kwargs)
