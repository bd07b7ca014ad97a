"""GPU tests (pytest -m gpu) of the multi-rank path as far as one GPU allows.

Instances are independent, so the only exchange between ranks is the gather
of the per-step results (nmpc_amd/sharding.py).  Two things are shown on
device memory here, each in a fresh child process:

  * the gather over RCCL itself (backend "nccl", one rank): the results of a
    real 4 096-instance solve go through sharding.gather_results and through
    three overlapped ResultGather rounds as CUDA tensors and come back equal;
  * bench.py's N > 1 branch end to end: two ranks started by
    torch.distributed.run share the one GPU (NMPC_BENCH_REHEARSE=1, gather
    staged over gloo because RCCL refuses two ranks on one device), each owns
    2 048 instances of its own seed, and the gathered results equal the CPU
    oracle's, rank range by rank range.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import assert_same

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_RCCL_CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
from nmpc_amd import sharding, synthetic as S
from nmpc_amd.batch import Batch

os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = %(port)r
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"

B, N = 4096, 20
d = S.make_batch(B, N, seed=777)
bt = Batch(B, N)
bt.setup(**S.setup_args())
bt.initialise_horizon()
bt.set_wind_velocity(d["wind"])
bt.set_reference(d["refs"][:, :N + 1])
bt.solve(d["meas"][:, 0])
u = torch.zeros((B, 3), dtype=torch.float64, device=dev)
st = torch.zeros(B, dtype=torch.int32, device=dev)
it = torch.zeros(B, dtype=torch.int32, device=dev)
bt.get_controls_into(u, st, it)
bt.synchronize()
assert int((st == 0).sum()) > B // 2 and float(u.abs().sum()) > 0

gu, gs, gi = sharding.gather_results(u, st, it)                 # counts exchanged first
assert gu.is_cuda and torch.equal(gu, u) and torch.equal(gs, st) and torch.equal(gi, it)
ku, ks, ki = sharding.gather_results(u, st, it, counts=[B])
assert torch.equal(ku, u) and torch.equal(ks, st) and torch.equal(ki, it)

g = sharding.ResultGather([B], dev)
assert not g.staged
for k in range(3):                                              # three rounds in flight one after the other
    g.start(u + k, st, it + k)
lu, ls, li = g.finish()
assert lu.is_cuda and torch.equal(lu, u + 2) and torch.equal(ls, st) and torch.equal(li, it + 2)
torch.cuda.synchronize()

loaded = [l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l]
assert loaded, "RCCL is not mapped into the process"
bt.close()
dist.destroy_process_group()
print("RCCL-GATHER-OK", sorted(set(os.path.basename(p) for p in loaded))[0])
"""


def test_result_gather_over_rccl_on_device_memory():
    code = _RCCL_CHILD % dict(root=ROOT, port=str(_free_port()))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "RCCL-GATHER-OK" in r.stdout, r.stdout[-1500:]


def test_two_rank_bench_rehearsal_on_one_gpu(tmp_path, port_built):
    import portdriver as P
    from nmpc_amd import synthetic as S
    B, N, world = 2048, 100, 2
    dump = str(tmp_path / "gathered.npz")
    env = dict(os.environ, NMPC_BENCH_REHEARSE="1", NMPC_BENCH_DUMP=dump)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--instances", str(B),
           "--steps", "3", "--warmup", "1", "--no-peaks"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]           # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["scaling"] == "weak"
    assert line["config"]["total_instances"] == world * B
    assert line["config"]["instances_per_gpu"] == B
    assert line["value"] > 0 and line["cpu_baseline"] is None
    g = np.load(dump)
    assert g["u"].shape == (world * B, 3)
    cfg = S.setup_args()
    for rank in range(world):                           # every rank drew its instances from seed + rank
        d = S.make_batch(B, N, seed=S.DEFAULT_SEED + rank)
        eu, est, eit, _ = P.batch_cold_solve(
            N, cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"],
            cfg["lower_control_bound"], cfg["upper_control_bound"], d["refs"][:, :N + 1],
            d["meas"][:, 0], d["wind"], threads=8)
        lo, hi = rank * B, (rank + 1) * B
        assert_same(g["status"][lo:hi], est, "status, rank %d" % rank)
        assert_same(g["iterations"][lo:hi], eit, "QP iterations, rank %d" % rank)
        assert_same(g["u"][lo:hi], eu, "controls, rank %d" % rank)
    assert (g["status"] == 0).sum() > world * B // 2
