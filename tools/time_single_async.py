"""One controller through the batch interface with device-resident inputs: a control step
with one synchronisation at its end, against the legacy interface's one per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nmpc_amd import synthetic as S
from nmpc_amd.batch import Batch

N, W = 100, 300
dev = torch.device("cuda:0")
d = S.make_batch(1, N, extra_steps=W + 2, seed=S.DEFAULT_SEED)
refs = torch.from_numpy(np.ascontiguousarray(d["refs"])).to(dev)
meas = torch.from_numpy(np.ascontiguousarray(d["meas"])).to(dev)
wind = torch.from_numpy(d["wind"]).to(dev)
u = torch.zeros((1, 3), dtype=torch.float64, device=dev)
st = torch.zeros(1, dtype=torch.int32, device=dev); it = torch.zeros(1, dtype=torch.int32, device=dev)
bt = Batch(1, N, stream=torch.cuda.current_stream().cuda_stream)
bt.setup(**S.setup_args())
bt.initialise_horizon(); bt.set_wind_velocity(wind); bt.set_reference(refs[:, :N + 1].contiguous())
bt.solve(meas[:, 0].contiguous()); bt.get_controls_into(u, st, it)
torch.cuda.synchronize()
cols = [refs[:, N + 1 + k].contiguous() for k in range(W)]
ms = [meas[:, k + 1].contiguous() for k in range(W)]
lat = []
for k in range(W):
    t0 = time.perf_counter()
    bt.update_horizon(cols[k]); bt.prepare(); bt.set_wind_velocity(wind)
    bt.solve(ms[k]); bt.get_controls_into(u, st, it)
    torch.cuda.synchronize()
    lat.append((time.perf_counter() - t0) * 1e3)
lat = sorted(lat[50:])
print("one synchronisation per control step: p50 %.4f ms p99 %.4f ms" % (lat[len(lat) // 2], lat[int(0.99 * len(lat))]))
