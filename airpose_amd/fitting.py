"""AirPose+ fitting loop on MI355X (BASELINE config 5): mirror of the optimisation of
copenet_real_data/scripts/bundle_adj.py:262-401 -- 300 Adam steps on the VPoser latent, the per-view root pose and the
shared shape against 2-D joint detections in both views.  All compute in libairpose_hip.so (ap_fit_run: hand-written
adjoints, no autograd); there is no CPU path."""
import ctypes

import torch

from . import _native as N

W_VPOSER, W_TEMPORAL, GM_SIGMA, LR, SWITCH_ITER = 0.05, 1.0, 30.0, 0.01, 100      # bundle_adj.py:134,243-245,279-295


class AirPosePlusFitter:
    """vposer: dict with the decoder's Linear layers `w1,b1,w2,b2,w3,b3` (VPoser V02_05 `decoder_net.{0,3,5}`);
    body: an airpose_amd.smplx.SMPLX (rest joints and their shape directions come from its native handle)."""

    def __init__(self, vposer, body, device=None):
        N.require_gpu()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.body = body
        f = lambda t: torch.as_tensor(t, dtype=torch.float32).contiguous().cpu()
        w = [f(vposer[k]) for k in ("w1", "b1", "w2", "b2", "w3", "b3")]
        if w[0].shape != (512, 32) or w[2].shape != (512, 512) or w[4].shape != (126, 512):
            raise RuntimeError("VPoser decoder layers must be 512x32, 512x512, 126x512")
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            N.check(N.lib().ap_fit_create(ctypes.byref(self._h), body._native(self.device),
                                          *[ctypes.c_void_p(t.data_ptr()) for t in w], self.device.index or 0), "ap_fit_create")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                N.lib().ap_fit_destroy(h)
            except Exception:
                pass

    def run(self, state, j2d, robust, intr, extr, n_iters=300, first_iter=0, switch_iter=SWITCH_ITER, lr=LR, sigma=GM_SIGMA,
            w_vposer=W_VPOSER, w_temporal=W_TEMPORAL, want_loss=False, want_grad=False):
        """state: dict z (L,32), phi0/phi1 (L,6), tau0/tau1 (L,3), beta (10,) -- returns the fitted copy (+ loss history
        (n_iters,4) = [2-D, temporal pose, temporal rigid, VPoser prior] and the last gradient dict on request)."""
        dev = self.device
        L = state["z"].shape[0]
        z = N.f32c(state["z"], dev).clone()
        phi = torch.stack([N.f32c(state["phi0"], dev), N.f32c(state["phi1"], dev)]).contiguous()
        tau = torch.stack([N.f32c(state["tau0"], dev), N.f32c(state["tau1"], dev)]).contiguous()
        beta = N.f32c(state["beta"], dev).clone()
        j2d, intr, extr = N.f32c(j2d, dev), N.f32c(intr, dev), N.f32c(extr, dev)
        if j2d.shape != (2, L, 2, 24, 3) or intr.shape != (2, 4) or extr.shape != (2, 3, 4):
            raise RuntimeError("j2d (2,L,2,24,3), intr (2,4), extr (2,3,4)")
        if z.shape != (L, 32) or phi.shape != (2, L, 6) or tau.shape != (2, L, 3) or beta.shape != (10,):
            raise RuntimeError("state: z (L,32), phi0/phi1 (L,6), tau0/tau1 (L,3), beta (10,)")
        rob = torch.as_tensor(robust).to(torch.int32).cpu().contiguous()
        if rob.shape != (L,):
            raise RuntimeError("robust must hold one flag per frame")
        if L < 1 or n_iters < 1:
            raise RuntimeError("need at least one frame and one iteration")
        hist = torch.zeros(n_iters, L, 4, device=dev) if want_loss else None
        grad = torch.zeros(L * 32 + 2 * L * 9 + 10, device=dev) if want_grad else None
        with torch.cuda.device(dev):
            N.check(N.lib().ap_fit_run(self._h, L, N.dptr(z), N.dptr(phi), N.dptr(tau), N.dptr(beta), N.dptr(j2d),
                                       ctypes.c_void_p(rob.data_ptr()), N.dptr(intr), N.dptr(extr), int(first_iter), int(n_iters),
                                       int(switch_iter), float(lr), float(sigma), float(w_vposer), float(w_temporal),
                                       N.dptr(hist) if hist is not None else None, N.dptr(grad) if grad is not None else None,
                                       N.stream_ptr(dev)), "ap_fit_run")
        out = {"z": z, "phi0": phi[0], "phi1": phi[1], "tau0": tau[0], "tau1": tau[1], "beta": beta}
        res = [out]
        if want_loss:
            lh = hist.sum(1)
            lh[:, 3] = 0.0      # (the prior is a function of z alone; reported by the caller if needed)
            res.append(lh)
        if want_grad:
            nz, nphi = L * 32, 2 * L * 6
            g = {"z": grad[:nz].view(L, 32), "phi0": grad[nz:nz + L * 6].view(L, 6), "phi1": grad[nz + L * 6:nz + nphi].view(L, 6),
                 "tau0": grad[nz + nphi:nz + nphi + L * 3].view(L, 3), "tau1": grad[nz + nphi + L * 3:nz + nphi + L * 6].view(L, 3),
                 "beta": grad[nz + nphi + L * 6:]}
            res.append(g)
        return res[0] if len(res) == 1 else tuple(res)


def synthetic_fit_problem(frames, seed, device):
    """Seeded stand-in for the gated inputs of the fitting loop (VPoser V02_05 weights, OpenPose / AlphaPose detections), for
    LATENCY measurements: a random-initialised decoder (nn.Linear init), an initial state around the upright pose and 2-D
    detections scattered around the projection of a body 8 m in front of each camera.  The loop runs a fixed number of
    iterations whatever the data; tests that need observations consistent with a ground-truth motion build them with the
    oracle's generator instead.  Returns (vposer, state, data) with data = j2d (2,L,2,24,3), robust (L,), intr (2,4), extr (2,3,4)."""
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=g)
    uni = lambda *s: torch.rand(*s, generator=g)
    L = frames

    def lin(o, i):
        b = 1.0 / i ** 0.5
        return (uni(o, i) * 2 - 1) * b, (uni(o) * 2 - 1) * b
    vp = {}
    vp["w1"], vp["b1"] = lin(512, 32)
    vp["w2"], vp["b2"] = lin(512, 512)
    vp["w3"], vp["b3"] = lin(126, 512)
    t = torch.linspace(0, 1, L).unsqueeze(1)
    st = {"z": 0.8 * rnd(1, 32) + 0.3 * rnd(L, 32), "beta": torch.zeros(10)}
    for v in (0, 1):
        st["phi%d" % v] = torch.tensor([[1.0, 0, 0, 0, -1.0, 0]]) + 0.15 * rnd(1, 6) + 0.05 * t * rnd(1, 6) + 0.05 * rnd(L, 6)
        st["tau%d" % v] = torch.tensor([[0.0, 0.2, 8.0]]) + 0.3 * rnd(1, 3) + 0.4 * t * rnd(1, 3) + 0.2 * rnd(L, 3)
    intr = torch.tensor([[1475.0, 1475.0, 960.0, 540.0], [1470.0, 1480.0, 950.0, 545.0]])
    extr = torch.eye(4)[:3].unsqueeze(0).repeat(2, 1, 1).contiguous()
    j2d = torch.zeros(2, L, 2, 24, 3)
    for v in (0, 1):
        centre = torch.stack([intr[v, 2] + 40.0 * rnd(L, 1), intr[v, 3] + 40.0 * rnd(L, 1)], -1)      # (L,1,2): the body's image
        limbs = 90.0 * rnd(1, 24, 2)                                                               # a fixed 24-joint layout around it
        for det in (0, 1):
            j2d[v, :, det, :, :2] = centre + limbs + 3.0 * rnd(L, 24, 2)
            j2d[v, :, det, :, 2] = uni(L, 24)
    data = {"j2d": j2d, "robust": uni(L) > 0.1, "intr": intr, "extr": extr}
    dev = torch.device(device)
    return vp, {k: v.to(dev) for k, v in st.items()}, {k: (v.to(dev) if v.is_floating_point() else v) for k, v in data.items()}
