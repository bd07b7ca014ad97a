"""CPU test of the N>1 path: two gloo ranks each own a contiguous instance
range, produce their results (here with the CPU oracle standing in for the
GPU) and all-gather them; the gathered result must equal the one-process
result.  No data-path collective exists besides this gather."""
import os
import socket

import numpy as np
import pytest

from nmpc_amd import sharding
from nmpc_amd import synthetic as S


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _solve_range(lo, hi, total, N, seed):
    import portdriver as P
    d = S.make_batch(total, N, seed=seed, turn_start=(0.3, 3.0))   # same inputs on every rank
    cfg = S.setup_args()
    return P.batch_cold_solve(
        N, cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"],
        cfg["lower_control_bound"], cfg["upper_control_bound"], d["refs"][lo:hi],
        d["meas"][lo:hi, 0], d["wind"][lo:hi], threads=2)[:3]


def _worker(rank, world, port, total, N, seed, out_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.instance_range(total, world, rank)
    u, st, it = _solve_range(lo, hi, total, N, seed)
    gu, gs, gi = sharding.gather_results(torch.from_numpy(u), torch.from_numpy(st),
                                         torch.from_numpy(it))
    # the same with the partition known up front (no count exchange per call)
    counts = [b - a for a, b in (sharding.instance_range(total, world, r) for r in range(world))]
    assert counts == sharding.gather_counts(hi - lo, torch.device("cpu"))
    ku, ks, ki = sharding.gather_results(torch.from_numpy(u), torch.from_numpy(st),
                                         torch.from_numpy(it), counts=counts)
    assert torch.equal(ku, gu) and torch.equal(ks, gs) and torch.equal(ki, gi)
    # the control-loop form: three steps in flight one after the other, buffers reused
    g = sharding.ResultGather(counts, "cpu")
    for k in range(3):
        g.start(torch.from_numpy(u + k), torch.from_numpy(st), torch.from_numpy(it + k))
    lu, ls, li = g.finish()
    assert torch.equal(lu, gu + 2) and torch.equal(ls, gs) and torch.equal(li, gi + 2)
    dist.barrier()
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), u=gu.numpy(), st=gs.numpy(), it=gi.numpy())
    dist.destroy_process_group()


def test_two_rank_gather(tmp_path, port_built):
    import torch.multiprocessing as mp
    total, N, seed = 7, 20, 31        # uneven split: 4 + 3
    mp.spawn(_worker, args=(2, _free_port(), total, N, seed, str(tmp_path)), nprocs=2, join=True)
    g = np.load(tmp_path / "gathered.npz")
    u, st, it = _solve_range(0, total, total, N, seed)
    assert np.array_equal(g["u"], u)
    assert np.array_equal(g["st"], st) and np.array_equal(g["it"], it)
