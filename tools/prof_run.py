"""Profiling workload: the bench step (4096 instances, horizon 100, cold solve)
run a few times without the CPU baseline; used under rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nmpc_amd import synthetic as S
from nmpc_amd.batch import Batch

B = int(os.environ.get("PROF_B", "4096")); N = int(os.environ.get("PROF_N", "100"))
steps = int(os.environ.get("PROF_STEPS", "3"))
dev = torch.device("cuda:0")
d = S.make_batch(B, N, seed=S.DEFAULT_SEED)
refs = torch.from_numpy(np.ascontiguousarray(d["refs"][:, :N + 1])).to(dev)
meas = torch.from_numpy(np.ascontiguousarray(d["meas"][:, 0])).to(dev)
wind = torch.from_numpy(d["wind"]).to(dev)
u = torch.zeros((B, 3), dtype=torch.float64, device=dev)
st = torch.zeros(B, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
bt = Batch(B, N, stream=torch.cuda.current_stream().cuda_stream)
bt.setup(**S.setup_args())
for _ in range(steps):
    bt.reset(); bt.initialise_horizon(); bt.set_wind_velocity(wind); bt.set_reference(refs)
    bt.solve(meas); bt.get_controls_into(u, st, it)
torch.cuda.synchronize()
print("done", float(u.sum()), int(it.sum()))
