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
    return (full[:, 0:3].contiguous(), words[:, 6].contiguous(), words[:, 7].contiguous())
