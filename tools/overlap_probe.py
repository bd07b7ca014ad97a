"""Do the kernels of two batches on two streams overlap?  Cold solves of one batch of B
instances against two batches of B/2 (and four of B/4) issued back to back."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nmpc_amd import synthetic as S
from nmpc_amd.batch import Batch

B, N, steps = int(os.environ.get("PB", "4096")), 100, 10
dev = torch.device("cuda:0")
d = S.make_batch(B, N, seed=S.DEFAULT_SEED)
for parts in (1, 2, 4):
    n = B // parts
    bts, ins, outs = [], [], []
    for i in range(parts):
        sl = slice(i * n, (i + 1) * n)
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            refs = torch.from_numpy(np.ascontiguousarray(d["refs"][sl, :N + 1])).to(dev)
            meas = torch.from_numpy(np.ascontiguousarray(d["meas"][sl, 0])).to(dev)
            wind = torch.from_numpy(np.ascontiguousarray(d["wind"][sl])).to(dev)
        bt = Batch(n, N, stream=st.cuda_stream)
        bt.setup(**S.setup_args())
        bts.append(bt); ins.append((refs, meas, wind))
        outs.append((torch.zeros((n, 3), dtype=torch.float64, device=dev),
                     torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)))
    torch.cuda.synchronize()

    def step():
        for bt, (refs, meas, wind), o in zip(bts, ins, outs):
            bt.reset(); bt.initialise_horizon(); bt.set_wind_velocity(wind); bt.set_reference(refs)
            bt.solve(meas); bt.get_controls_into(*o)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print("%d batch(es) of %d: %.3f ms per step of %d instances, %.0f solves/s" % (parts, n, ms, B, B / ms * 1e3))
    for bt in bts:
        bt.close()
