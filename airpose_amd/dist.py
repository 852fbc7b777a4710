"""Multi-GPU host logic (one process per GPU, torch.distributed: "nccl" = RCCL over xGMI on the box,
"gloo" in the CPU tests).

Default sharding needs no collective: pairs are independent, rank r owns a contiguous block of whole
pairs.  A collective exists only in VIEW-SPLIT mode -- view 0 on rank 2k, view 1 on rank
2k+1, the on-drone topology of the reference (README.md:238-241: step1/step2/step3 with the partner's
state exchanged between steps).  What moves per exchange is the partner's [art_pose(126) | shape(10)]
= 136 floats = 544 B per sample (model_copenet.py:185,192); it is latency-bound, so it is a 2-rank
all_gather on a pair group (the direct xGMI link), never a ring over all ranks.
"""
import torch
import torch.distributed as dist


def shard_pairs(n_global, rank, world):
    """Contiguous [start, stop) block of pairs owned by `rank` (remainder spread over the first ranks)."""
    q, r = divmod(n_global, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def make_pair_groups(world, timeout_s=None):
    """One 2-rank group per (2k, 2k+1); every rank must call this (new_group is collective).  timeout_s bounds every
    collective on these groups (a lost partner then raises instead of hanging the caller)."""
    if world % 2:
        raise ValueError("view-split mode needs an even number of ranks")
    kw = {}
    if timeout_s:
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=timeout_s)
    return [dist.new_group([2 * k, 2 * k + 1], **kw) for k in range(world // 2)]


def pack_wire(pose, betas):
    """The reference's published result vector (copenet_real/scripts/copenet_rosViz.py:83-85):
    145 floats = beta(10) | trans(3) | 6D pose (132)."""
    return torch.cat([betas, pose[:, :3], pose[:, 3:]], dim=1)


def unpack_wire(msg):
    return torch.cat([msg[:, 10:13], msg[:, 13:]], dim=1), msg[:, :10]


class ViewSplitIEF(object):
    """IEF loop of model_copenet.py:144-157 with the two views on two ranks.

    step_fn(xf, bb, pose (B,135), betas (B,10), partner (B,136)) -> (pose, betas) is one forward_reg
    evaluation for this rank's view: airpose_amd.copenet_model.copenet.regressor_step on the GPU.
    """

    def __init__(self, step_fn, pair_group, pair_ranks, split_step=None):
        """split_step = (feat_part(xf) -> hfeat, step_local(hfeat, bb, pose, betas) -> partial, step_finish(partial, pose, betas,
        partner) -> (pose, betas)): the step in its partner-independent and partner-dependent halves (copenet.regressor_feat_part /
        _step_local / _step_finish).  With it ``run`` hides the exchange behind the local half."""
        self.step_fn, self.group, self.ranks = step_fn, pair_group, tuple(pair_ranks)
        self.split_step = split_step
        self.me = self.ranks.index(dist.get_rank())

        self.n_exchanges = 0                                              # collectives issued so far (bench / tests)
        # RCCL ("nccl") moves device buffers directly over xGMI; gloo (CPU tests, and the 2-process test that shares
        # one GPU) takes host tensors, so the 544 B/sample message is staged through the host there
        self._host_staged = dist.get_backend(pair_group) == "gloo"
        self._buf_key, self._mine, self._both, self._dev = None, None, None, None

    def _buffers(self, B, device):
        """Preallocated send row and (2, B, 136) gather buffer, cached per (B, device): no allocation per exchange."""
        key = (int(B), str(device))
        if self._buf_key != key:
            self._mine = torch.empty(B, 136, dtype=torch.float32, device=device)
            self._both = torch.empty(2, B, 136, dtype=torch.float32, device=device)
            self._buf_key = key
        return self._mine, self._both

    def exchange_start(self, pose, betas):
        """Issue the 2-rank all-gather of [art_pose | shape] into the preallocated (2, B, 136) buffer
        (all_gather_into_tensor: one flat receive buffer, no per-rank list copies) and return its work handle;
        the caller may run this view's partner-independent kernels before exchange_wait()."""
        host = self._host_staged and pose.is_cuda
        mine, both = self._buffers(pose.shape[0], torch.device("cpu") if host else pose.device)
        mine[:, :126].copy_(pose[:, 9:135], non_blocking=not host)        # art_pose | shape
        mine[:, 126:].copy_(betas, non_blocking=not host)
        self._dev = pose.device
        self.n_exchanges += 1
        return dist.all_gather_into_tensor(both.view(2 * both.shape[1], 136), mine, group=self.group, async_op=True)

    def exchange_wait(self, work):
        """The partner's (B, 136) rows as a tensor the CALLER owns: a copy, never a view of the cached gather buffer (the
        next exchange_start overwrites that buffer in place; 544 B per sample)."""
        work.wait()                                                       # RCCL: orders the current stream behind the collective
        part = self._both[1 - self.me]
        if self._host_staged:
            # host-staged (gloo) message: a synchronous copy out of the reused pageable buffer -- an asynchronous H2D from it
            # could still be reading when the next all_gather writes it
            return part.to(self._dev, non_blocking=False) if self._dev.type != "cpu" else part.clone()
        return part.clone()

    def exchange(self, pose, betas):
        return self.exchange_wait(self.exchange_start(pose, betas))

    def run(self, xf, bb, init_position, init_theta, init_shape, iters=3, shared_init=False, overlap=None):
        """overlap (default: on when the step comes in halves): the exchange of iteration k is issued as soon as the state of
        iteration k - 1 exists, the 2196 partner-independent columns of the step run while the 544 B / sample are on the wire,
        and only the 136 partner columns + the residual add wait for it (model_copenet.py:185-199; fc1 -> fc2 -> dec is linear).

        shared_init=True is an opt-in promise that both ranks were handed the SAME init_theta / init_shape (the
        model's mean parameters, model_copenet.py:121-136): the partner's initial [art_pose | shape] is then this
        rank's own and iteration 1 needs no exchange: iters - 1 collectives per forward.  The default
        (False) exchanges before every iteration, which is always correct, also with per-view caller state
        (init_theta0 != init_theta1)."""
        B = xf.shape[0]
        theta = init_theta[:, :132].expand(B, -1)
        pose = torch.cat([init_position, theta], dim=1).contiguous()
        betas = init_shape.expand(B, -1).contiguous()
        if overlap is None:
            overlap = self.split_step is not None
        if overlap:
            if self.split_step is None:
                raise ValueError("overlap=True needs split_step")
            feat_part, step_local, step_finish = self.split_step
            hfeat = feat_part(xf)                                         # constant over the iterations
            for it in range(int(iters)):
                work = None if (it == 0 and shared_init) else self.exchange_start(pose, betas)
                partial = step_local(hfeat, bb, pose, betas)              # ... while the partner's rows travel
                partner = torch.cat([pose[:, 9:], betas], dim=1).contiguous() if work is None else self.exchange_wait(work)
                pose, betas = step_finish(partial, pose, betas, partner)
            return pose, betas
        for it in range(int(iters)):
            if it == 0 and shared_init:
                partner = torch.cat([pose[:, 9:], betas], dim=1).contiguous()
            else:
                partner = self.exchange(pose, betas)
            pose, betas = self.step_fn(xf, bb, pose, betas, partner)
        return pose, betas
