"""Host mirror of the inference branch of ``copenet_twoview.fwd_pass_and_loss``
(copenet/src/copenet/copenet_twoview.py:166-223, 236-279, 307-350): batch dict in, the reference's
test-mode output dict out.  Python only orchestrates: TWO C-ABI calls per forward -- ap_copenet_fwd and
ap_smplx_fwd_twoview (translation un-scale in place on pred_pose, rot6d, SMPL-X for both views, root transform,
projection with the camera centres read from the intrinsics, optionally the test-mode input meshes) -- and no
torch kernel in between.

``submit`` is the serving form of the same forward: the trunk passes of batch i+1 queue directly behind those of batch i while
the IEF loop and the SMPL-X stage of batch i (0.12 ms of latency-bound and HBM-bound kernels behind a 5.5 ms MFMA-heavy trunk, plus
the cross-queue hops around them) run on a second stream.  Same kernels, same results; what changes is when the outputs are
ready (``Pending.wait``)."""
import torch

from . import _native as N

TRANS_SCALE = 0.05               # copenet_twoview.py:199
FOCAL_LENGTH = (1475.0, 1475.0)  # copenet/src/copenet/constants.py:7
DEPTH = 3                        # submit(): batches in flight (feature slots); the host blocks on the oldest when all are taken


class Pending(object):
    """Result of ``TwoViewInference.submit``: ``out`` (the dict ``__call__`` returns) is being produced on the pipeline's second
    stream.  ``wait()`` makes a stream (default: the current one) wait for it, ``synchronize()`` blocks the host; reading ``out``
    before either is a race, exactly as with any tensor produced on another stream."""

    def __init__(self, out, done, stream, model=None, slot=None):
        self.out, self._done, self._stream, self._model, self._slot = out, done, stream, model, slot

    def wait(self, stream=None):
        stream = stream if stream is not None else torch.cuda.current_stream(self._stream.device)
        stream.wait_event(self._done)
        # the outputs were allocated on the pipeline's stream: tell the caching allocator that ``stream`` uses them too, or a block
        # freed by the consumer could be handed to the next tail while kernels of ``stream`` still read it
        for t in (self.out.values() if isinstance(self.out, dict) else self.out):
            if torch.is_tensor(t):
                t.record_stream(stream)
        return self.out

    def synchronize(self):
        """Block the host until THIS batch is done -- the batches submitted after it keep running.  precision="f16": raises
        RangeError if a stored activation of this or an earlier batch's trunk passes left the fp16 range: ``submit`` has every pass
        stream snapshot its range word behind the batch's last kernel there (ap_net_range_mark_next: a later batch cannot leak
        into it), and the slot is read here, behind the batch's own event, without touching any stream (the deferred sentinel
        alone would only speak up at the next forward -- never, for the last batch of a run)."""
        self._done.synchronize()
        if self._model is not None and self._slot is not None and getattr(self._model, "precision", None) == "f16":
            self._model.range_slot(self._slot)
        return self.wait()                                   # (the event is complete: only the allocator bookkeeping of wait() remains)


class Outputs(dict):
    """The output dict of ``TwoViewInference.__call__`` for fp16 storage: reading an entry for the first time waits for the forward's
    event and raises RangeError if a stored activation of that forward left the fp16 range (a result must never be read out of a
    forward that overflowed; the wait is the one a reader of the results needs anyway).  Afterwards a plain dict."""

    def __init__(self, data, pipe, slot, done):
        super().__init__(data)
        self._chk = (pipe, slot, done)

    def _settle(self):
        chk, self._chk = self._chk, None
        if chk is not None:
            pipe, slot, done = chk
            done.synchronize()
            pipe._unread[:] = [e for e in pipe._unread if e[1] is not done]
            pipe.model.range_slot(slot)

    def __getitem__(self, k):
        self._settle()
        return super().__getitem__(k)

    def get(self, k, default=None):
        self._settle()
        return super().get(k, default)

    def items(self):
        self._settle()
        return super().items()

    def values(self):
        self._settle()
        return super().values()

    def pop(self, *a):
        self._settle()
        return super().pop(*a)


class TwoViewInference(object):
    def __init__(self, model, smplx, iters=3, focal_length=FOCAL_LENGTH):
        self.model, self.smplx, self.iters, self.focal_length = model, smplx, iters, focal_length
        self._pos = {}
        self._pl = {}            # submit(): per (B, device) the second stream, two feature slots and their events

    def init_position(self, B, device):
        """[0,0,10] * 0.05 (copenet_twoview.py:184-185,201-203), cached per (B, device)."""
        key = (B, device)
        if key not in self._pos:
            self._pos[key] = (torch.tensor([0.0, 0.0, 10.0], device=device).expand(B, -1).clone() * TRANS_SCALE)
        return self._pos[key]

    def _behind_submits(self, dev):
        """The stream-ordered calls share the handles' workspaces (regressor, SMPL-X) with the tails of earlier ``submit`` calls,
        which run on the pipeline's own stream: order the current stream behind the youngest of them."""
        for (_, d), st in self._pl.items():
            if d == dev and st["n"] and st["busy"][(st["n"] - 1) % DEPTH]:
                torch.cuda.current_stream(dev).wait_event(st["done"][(st["n"] - 1) % DEPTH])

    def forward_net(self, im0, im1, bb0, bb1):
        B, dev = im0.shape[0], im0.device
        self._behind_submits(dev)
        pos = self.init_position(B, dev)
        return self.model(x0=im0, x1=im1, bb0=bb0, bb1=bb1, init_position0=pos, init_position1=pos, iters=self.iters)

    CALL_SLOTS = (N.AP_RANGE_SLOTS - 2, N.AP_RANGE_SLOTS - 1)   # range-flag snapshot slots of __call__, alternating (submit uses 0 .. DEPTH - 1)

    def __call__(self, batch, want_rotmat=True, want_angles=False, want_input_mesh=False, check_range=True):
        """The reference's stream-ordered forward: returns at once, the outputs are complete in the order of the current stream.
        precision="f16" and check_range (default): the forward snapshots the handle's range words behind its OWN trunk passes, and
        the returned dict checks that snapshot the first time an entry is read (``Outputs``: it waits for the forward's event --
        which a caller reading results does anyway -- and raises RangeError if a stored activation of THIS forward left the fp16
        range), so a one-shot caller never reads garbage out of an AP_OK; a loop that never looks at a result is told by a later
        call (the snapshots of unread forwards are read as soon as they are complete; the host runs at most two such forwards
        ahead).  check_range="sync": wait for the stream and check before returning; check_range=False: the handle's deferred
        sentinel only."""
        im0, im1 = batch["im0"], batch["im1"]
        dev = im0.device
        f16 = bool(check_range) and getattr(self.model, "precision", None) == "f16"
        pend = self.__dict__.setdefault("_unread", [])       # forwards nobody has looked at yet: (slot, event), oldest first
        while pend and (pend[0][1].query() or len(pend) >= len(self.CALL_SLOTS)):
            slot0, ev0 = pend.pop(0)                         # complete by now -- or its slot comes round: settle it (the host is then
            ev0.synchronize()                                # two forwards ahead of the GPU; nothing idles)
            self.model.range_slot(slot0)
        if f16:
            self._call_no = getattr(self, "_call_no", 0) + 1
            slot = self.CALL_SLOTS[self._call_no % len(self.CALL_SLOTS)]
            self.model.range_mark_next(slot)
        p0, b0, p1, b1 = self.forward_net(im0, im1, batch["bb0"], batch["bb1"])
        out = self._tail(p0, b0, p1, b1, batch, want_rotmat, want_angles, want_input_mesh)
        if not f16:
            return out
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(dev))
        if check_range == "sync":
            done.synchronize()
            self.model.range_slot(slot)
            return out
        pend.append((slot, done))
        return Outputs(out, self, slot, done)

    def submit_net(self, im0, im1, bb0, bb1):
        """``forward_net`` issued as ``submit`` issues the whole forward: Pending.out = (pred_pose0, pred_betas0, pred_pose1,
        pred_betas1)."""
        return self.submit({"im0": im0, "im1": im1, "bb0": bb0, "bb1": bb1}, net_only=True)

    def submit(self, batch, want_rotmat=True, want_angles=False, want_input_mesh=False, net_only=False):
        """The forward of ``__call__`` for a serving loop: the inputs are taken in the order of the CURRENT stream; trunk (both
        views, model_copenet.py:140-141) on the model's two pass streams, IEF loop (:144-157) + SMPL-X stage
        (copenet_twoview.py:222-257,307-317) on the pipeline's own stream, which is the one the passes join into
        (ap_trunk_fwd_twoview_async).  Returns at once with a ``Pending``; the current stream is not made to wait for anything, so
        the passes of the next ``submit`` queue directly behind this one's and its tail runs under them.  At most DEPTH = 3 batches are
        in flight (the next ``submit`` blocks the host until the oldest has finished); one feature buffer per batch in flight.  The inputs of
        a batch are referenced until its slot comes round again; do not overwrite them in place before ``Pending.wait`` /
        ``synchronize``.  ``model`` and ``smplx`` are used on the pipeline's stream while a batch is in flight: other callers of the
        same two objects go through this pipeline (``__call__`` / ``forward_net`` order themselves behind the submits) or wait for the
        last ``Pending`` first."""
        im0, im1 = batch["im0"], batch["im1"]
        B, dev = im0.shape[0], im0.device
        key = (B, dev)
        for (b2, d2), o in self._pl.items():                 # a batch of another size in flight: same handles, another stream
            if d2 == dev and b2 != B and o["n"] and o["busy"][(o["n"] - 1) % DEPTH]:
                o["done"][(o["n"] - 1) % DEPTH].synchronize()
        st = self._pl.get(key)
        if st is None:
            # default priority: a high-priority second stream was measured SLOWER than no overlap at all (42.1k against 43.2k
            # pairs/s at B = 256: the tail's workgroups displace the next trunk's stems); HIP offers no lower priority than 0
            st = self._pl[key] = {"side": torch.cuda.Stream(device=dev), "n": 0,
                                  "feat": [torch.empty(2, B, 2048, device=dev, dtype=torch.float32) for _ in range(DEPTH)],
                                  "done": [torch.cuda.Event() for _ in range(DEPTH)], "busy": [False] * DEPTH, "keep": [None] * DEPTH}
        side, slot = st["side"], st["n"] % DEPTH
        st["n"] += 1
        if st["busy"][slot]:                                 # DEPTH submits ago: its tail has read the slot's features and inputs
            st["done"][slot].synchronize()
        pos = self.init_position(B, dev)
        st["keep"][slot] = batch
        f16 = getattr(self.model, "precision", None) == "f16"
        if f16:                                              # each pass stream snapshots its range word behind this batch's last kernel
            self.model.range_mark_next(slot)
        feat = self.model.forward_feat_ext_twoview(im0, im1, out=st["feat"][slot], out_stream=side)
        with torch.cuda.stream(side):
            out = self.model.forward_ief(feat[0], feat[1], batch["bb0"], batch["bb1"], pos, pos, iters=self.iters)
            if not net_only:
                out = self._tail(*out, batch, want_rotmat, want_angles, want_input_mesh)
            st["done"][slot].record(side)
        st["busy"][slot] = True
        return Pending(out, st["done"][slot], side, self.model, slot if f16 else None)

    def _tail(self, p0, b0, p1, b1, batch, want_rotmat, want_angles, want_input_mesh):
        B, dev = p0.shape[0], p0.device
        # pred_pose0/1 and betas0/1 are the two halves of one (2,B,.) buffer: both views run as 2B bodies
        pose = p0._base if p0._base is not None and p0._base.shape == (2, B, 135) else torch.stack([p0, p1])
        betas = b0._base if b0._base is not None and b0._base.shape == (2, B, 10) else torch.stack([b0, b1])
        in_trans = None
        if want_input_mesh:
            # in_smpltrans after the reference's ``*= trans_scale`` ... ``/= trans_scale`` round trip (:199-203,216-218)
            key = ("in", B, dev)
            if key not in self._pos:
                t = self.init_position(B, dev) / TRANS_SCALE
                self._pos[key] = torch.stack([t, t]).contiguous()
            in_trans = self._pos[key]
        # ``pred_smpltrans /= trans_scale`` happens inside the native call, in place on pose (so the returned
        # pred_pose is un-scaled too, exactly as in the reference where pred_smpltrans is a view of pred_pose)
        o = self.smplx.forward_twoview(pose, betas, batch["intr0"], batch["intr1"], trans_scale=TRANS_SCALE,
                                       in_smpltrans=in_trans, focal_length=self.focal_length, want_rotmat=want_rotmat)
        out = {}
        for v in (0, 1):
            out["pred_pose%d" % v] = pose[v]
            out["pred_betas%d" % v] = betas[v]
            out["pred_smpltrans%d" % v] = pose[v, :, :3]
            out["pred_vertices_cam%d" % v] = o["vertices_cam"][v]
            out["pred_j3d_cam%d" % v] = o["j3d_cam"][v]
            out["pred_j2d_cam%d" % v] = o["j2d_cam"][v]
            if want_rotmat:
                out["pred_rotmat%d" % v] = o["rotmat"][v]
            if want_input_mesh:                            # :258-279, 330-331, 342-343
                out["pred_vertices_cam_in%d" % v] = o["vertices_cam_in"][v]
                out["in_smpltrans%d" % v] = in_trans[v]
        if want_angles:                                    # test-mode output of the caller, :323-324
            if not want_rotmat:
                raise ValueError("want_angles needs want_rotmat")
            from .geometry import rotation_matrix_to_angle_axis
            ang = rotation_matrix_to_angle_axis(o["rotmat"].reshape(-1, 3, 3)).view(2, B, 22, 3)
            out["pred_angles0"], out["pred_angles1"] = ang[0], ang[1]
        return out
