"""Instance partitioning for one-process-per-GPU runs.

Controller instances are independent, so rank r of `world` owns the contiguous
instance range instance_range(total, world, r); its per-instance state never
leaves its GPU.  The only exchange is the gather of the per-step results
(controls 3 x f64, status i32, iterations i32 per instance)."""


def instance_range(total, world, rank):
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_counts(count, device, group=None):
    """Instance counts of every rank (one small all-gather and a host read)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n = torch.tensor([int(count)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    return [int(c.item()) for c in counts]


def gather_results(controls, status, iterations, group=None, counts=None):
    """All-gather of per-rank results along the instance axis with
    torch.distributed (RCCL over xGMI on GPUs, gloo on CPU).  Inputs are torch
    tensors [b_r][3] f64, [b_r] i32, [b_r] i32 with possibly unequal b_r;
    returns the concatenation over ranks in rank order.

    counts: the b_r of every rank when the caller knows them (gather_counts
    once, or a fixed partition); without it every call exchanges them first,
    which waits for the device."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if counts is None:
        counts = gather_counts(controls.shape[0], controls.device, group)
    m = max(counts)
    # one packed 32-byte record per instance: u0 u1 u2 (f64), status, iterations (i32)
    rec = torch.zeros((m, 4), dtype=torch.float64, device=controls.device)
    b = controls.shape[0]
    rec[:b, 0:3] = controls
    tail = rec.view(torch.int32)        # [m][8]: words 6 and 7 are the fourth double
    tail[:b, 6] = status
    tail[:b, 7] = iterations
    if dist.get_backend(group) == "gloo" and rec.is_cuda:
        # gloo works on host memory: stage the records through it (rehearsals of the
        # multi-rank path on fewer GPUs than ranks; RCCL gathers device to device)
        host = torch.empty((world * m, 4), dtype=torch.float64)
        dist.all_gather_into_tensor(host, rec.cpu(), group=group)
        full = host.to(controls.device)
    else:
        full = torch.empty((world * m, 4), dtype=torch.float64, device=controls.device)
        dist.all_gather_into_tensor(full, rec, group=group)
    if any(c != m for c in counts):
        full = torch.cat([full[r * m:r * m + c] for r, c in enumerate(counts)], dim=0)
    words = full.view(torch.int32)
    # (records are float64 whatever the library's precision: exact for float32 controls,
    # which come back in the type they went in)
    return (full[:, 0:3].to(controls.dtype).contiguous(), words[:, 6].contiguous(), words[:, 7].contiguous())


class ResultGather:
    """The same gather for a control loop: buffers are allocated once, and the
    collective of step k runs beside the kernels of step k+1 (the next step needs
    nothing from it), so ranks do not wait for each other inside a step.

        g = ResultGather(counts, device)
        for every step:  ...solve...; g.start(controls, status, iterations)
        u, st, it = g.finish()            # the results of the last step started

    start() packs this rank's records and issues the all-gather without waiting
    for it; two sets of buffers alternate, and a set is reused only after its
    collective has completed.  finish() waits for the last collective."""

    def __init__(self, counts, device, group=None):
        import torch
        import torch.distributed as dist

        self.counts = [int(c) for c in counts]
        self.group = group
        self.world = dist.get_world_size(group)
        assert len(self.counts) == self.world
        self.m = max(self.counts)
        self.device = torch.device(device)
        self.staged = dist.get_backend(group) == "gloo" and self.device.type == "cuda"
        side = torch.device("cpu") if self.staged else self.device
        self.rec = [torch.zeros((self.m, 4), dtype=torch.float64, device=self.device) for _ in range(2)]
        self.full = [torch.empty((self.world * self.m, 4), dtype=torch.float64, device=side) for _ in range(2)]
        self.work = [None, None]
        self.turn = 0
        self.last = None

    def start(self, controls, status, iterations):
        import torch
        import torch.distributed as dist

        i = self.turn
        self.turn ^= 1
        if self.work[i] is not None:
            self.work[i].wait()
        rec = self.rec[i]
        b = controls.shape[0]
        rec[:b, 0:3] = controls
        words = rec.view(torch.int32)       # [m][8]: words 6 and 7 are the fourth double
        words[:b, 6] = status
        words[:b, 7] = iterations
        src = rec.cpu() if self.staged else rec
        self.work[i] = dist.all_gather_into_tensor(self.full[i], src, group=self.group, async_op=True)
        self.last = i
        self.dtype = controls.dtype

    def finish(self):
        import torch

        assert self.last is not None, "start() has not been called"
        i = self.last
        self.work[i].wait()
        self.work[i] = None
        full = self.full[i].to(self.device)
        if any(c != self.m for c in self.counts):
            full = torch.cat([full[r * self.m:r * self.m + c] for r, c in enumerate(self.counts)], dim=0)
        words = full.view(torch.int32)
        return (full[:, 0:3].to(self.dtype).contiguous(), words[:, 6].contiguous(), words[:, 7].contiguous())
