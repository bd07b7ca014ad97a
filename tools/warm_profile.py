"""Per-kernel times of warm control steps (feedback + horizon shift) after a cold solve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nmpc_amd import synthetic as S
from nmpc_amd.batch import Batch

B, N, W = int(os.environ.get("PROF_B", "4096")), int(os.environ.get("PROF_N", "100")), int(os.environ.get("WARM", "10"))
dev = torch.device("cuda:0")
d = S.make_batch(B, N, extra_steps=W, seed=S.DEFAULT_SEED)
refs = torch.from_numpy(np.ascontiguousarray(d["refs"])).to(dev)
meas = torch.from_numpy(np.ascontiguousarray(d["meas"])).to(dev)
wind = torch.from_numpy(d["wind"]).to(dev)
u = torch.zeros((B, 3), dtype=torch.float64, device=dev)
st = torch.zeros(B, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
bt = Batch(B, N, stream=torch.cuda.current_stream().cuda_stream)
bt.setup(**S.setup_args())
bt.initialise_horizon(); bt.set_wind_velocity(wind); bt.set_reference(refs[:, :N + 1].contiguous())
bt.solve(meas[:, 0].contiguous()); bt.get_controls_into(u, st, it)
bt.update_horizon(refs[:, N + 1].contiguous())
torch.cuda.synchronize()
print("cold: ok %.3f mean it %.2f" % (float((st == 0).float().mean()), float(it.float().mean())))
for k in range(1, W):
    bt.set_profiling(True)
    bt.solve(meas[:, k].contiguous()); bt.get_controls_into(u, st, it)
    bt.update_horizon(refs[:, N + 1 + k].contiguous())
    t = bt.kernel_times(); bt.set_profiling(False)
    hist = torch.bincount(it, minlength=6).tolist()
    print("warm %d: ok %.3f mean it %.2f hist %s  " % (k, float((st == 0).float().mean()), float(it.float().mean()), hist)
          + " ".join("%s %.2f/%d" % (n, ms, c) for n, (ms, c) in t.items() if c))
