"""Closed loop on the device, plain control steps against real-time-iteration steps:
B simulated vehicles (nmpc_batch_plant_*) follow their reference trajectories for a
number of control steps under B controllers, once with the reference's empty
preparation step and once with nmpc_batch_set_rti(1).  Reports, per mode, the
tracking error of the plant against the reference (position, velocity), the fraction
of solves that ended with NMPC_OK and the time per control step.

    python tools/rti_closed_loop.py [B] [steps] [seed]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from nmpc_amd import synthetic as S
from nmpc_amd.batch import Batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
seed = int(sys.argv[3]) if len(sys.argv) > 3 else S.DEFAULT_SEED
N, dt = 100, 1.0 / 50.0
dev = torch.device("cuda:0")
d = S.make_batch(B, N, extra_steps=steps, seed=seed)
refs = torch.from_numpy(np.ascontiguousarray(d["refs"])).to(dev)                 # [B][N+1+steps][16]
later = torch.from_numpy(np.ascontiguousarray(d["refs"][:, N + 1:].transpose(1, 0, 2))).to(dev)
wind = torch.from_numpy(d["wind"]).to(dev)
x0 = torch.from_numpy(np.ascontiguousarray(d["meas"][:, 0])).to(dev)

out = {}
for mode in ("plain", "rti"):
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    bt = Batch(B, N, stream=stream.cuda_stream)
    bt.setup(**S.setup_args())
    bt.set_rti(mode == "rti")
    bt.initialise_horizon()
    bt.set_wind_velocity(wind)
    bt.set_reference(refs[:, :N + 1].contiguous())
    bt.plant_set_state(x0)
    u = torch.zeros((B, 3), dtype=torch.float64, device=dev)
    st = torch.zeros(B, dtype=torch.int32, device=dev)
    it = torch.zeros(B, dtype=torch.int32, device=dev)
    x = torch.zeros((B, 13), dtype=torch.float64, device=dev)
    ok, iters, perr, verr = [], [], [], []
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(steps):
        bt.prepare()
        bt.solve_from_plant()
        bt.get_controls_into(u, st, it)
        bt.plant_integrate(dt)                  # the last good controls, the controllers' wind
        bt.update_horizon(later[k])
        bt._ck(bt.lib.nmpc_batch_plant_get_state(bt.h, x.data_ptr()))
        target = refs[:, k + 1, :13]
        perr.append((x[:, 0:3] - target[:, 0:3]).norm(dim=1).mean())
        verr.append((x[:, 3:6] - target[:, 3:6]).norm(dim=1).mean())
        ok.append((st == 0).double().mean())
        iters.append(it.double().mean())
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / steps * 1e3
    f = lambda v: [round(float(t), 4) for t in v]
    perr, verr, ok, iters = f(perr), f(verr), f(ok), f(iters)
    finite = bool(torch.isfinite(x).all())
    out[mode] = dict(ms_per_step=round(ms, 3), ok_frac=round(float(np.mean(ok)), 4),
                     mean_qp_iterations=round(float(np.mean(iters)), 3),
                     position_error_m=dict(mean=round(float(np.mean(perr)), 4), last_quarter=round(float(np.mean(perr[-steps // 4:])), 4)),
                     velocity_error_mps=dict(mean=round(float(np.mean(verr)), 4), last_quarter=round(float(np.mean(verr[-steps // 4:])), 4)),
                     plant_states_finite=finite)
    bt.close()
print(json.dumps(dict(instances=B, horizon=N, steps=steps, seed=seed, **out)))
