"""Wall-clock time of the bench step without per-kernel events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nmpc_amd import synthetic as S
from nmpc_amd.batch import Batch

B = int(os.environ.get("PROF_B", "4096")); N = int(os.environ.get("PROF_N", "100"))
steps = int(os.environ.get("PROF_STEPS", "10"))
dev = torch.device("cuda:0")
d = S.make_batch(B, N, seed=S.DEFAULT_SEED)
refs = torch.from_numpy(np.ascontiguousarray(d["refs"][:, :N + 1])).to(dev)
meas = torch.from_numpy(np.ascontiguousarray(d["meas"][:, 0])).to(dev)
wind = torch.from_numpy(d["wind"]).to(dev)
u = torch.zeros((B, 3), dtype=torch.float64, device=dev)
st = torch.zeros(B, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
bt = Batch(B, N, stream=torch.cuda.current_stream().cuda_stream)
bt.setup(**S.setup_args())
def step():
    bt.reset(); bt.initialise_horizon(); bt.set_wind_velocity(wind); bt.set_reference(refs)
    bt.solve(meas); bt.get_controls_into(u, st, it)
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print("ms_per_step %.4f solves/s %.1f checksum %r %d" % (dt * 1e3, B / dt, float(u.sum()), int(it.sum())))
