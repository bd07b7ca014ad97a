"""Latency of the single-controller interface (one instance per call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nmpc_amd import synthetic as S
from nmpc_amd.batch import Batch

for B in (1, 64, 512):
    N = 100
    d = S.make_batch(B, N, extra_steps=12, seed=5)
    bt = Batch(B, N)
    bt.setup(**S.setup_args()); bt.initialise_horizon(); bt.set_wind_velocity(d["wind"])
    bt.set_reference(d["refs"][:, :N + 1])
    ts = []
    for k in range(12):
        t0 = time.perf_counter()
        bt.solve(d["meas"][:, k]); u, st, it = bt.get_controls()
        ts.append(time.perf_counter() - t0)
        bt.update_horizon(d["refs"][:, N + 1 + k])
    print("B=%d: feedback + get_controls %.2f ms median (host buffers, synchronous)" % (B, 1e3 * sorted(ts)[6]))
    bt.close()
