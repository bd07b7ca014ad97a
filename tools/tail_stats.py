"""Step-length search statistics of warm control steps (library built with -DNMPC_TAIL_STATS)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nmpc_amd import synthetic as S
from nmpc_amd.batch import Batch

B, N, W = int(os.environ.get("TS_B", "4096")), int(os.environ.get("TS_N", "100")), int(os.environ.get("WARM", "4"))
dev = torch.device("cuda:0")
d = S.make_batch(B, N, extra_steps=W, seed=S.DEFAULT_SEED)
bt = Batch(B, N, stream=torch.cuda.current_stream().cuda_stream)
refs = torch.from_numpy(np.ascontiguousarray(d["refs"], bt.dtype)).to(dev)     # the library's precision
meas = torch.from_numpy(np.ascontiguousarray(d["meas"], bt.dtype)).to(dev)
wind = torch.from_numpy(np.ascontiguousarray(d["wind"], bt.dtype)).to(dev)
bt.setup(**S.setup_args())
bt.initialise_horizon(); bt.set_wind_velocity(wind); bt.set_reference(refs[:, :N + 1].contiguous())
out = (C.c_int * 16)()
bt.lib.nmpc_debug_tail_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
for k in range(W):
    bt.solve(meas[:, k].contiguous())
    bt.lib.nmpc_debug_tail_stats(bt.h, out)
    o = list(out)
    print("step %d: handed over per iteration %s, left open %s; searches %d rounds %d trials %d; set-up %.0f cycles/search, rounds %.0f cycles/round"
          % (k, o[0:5], o[8:13], o[5], o[6], o[7], o[13] * 256.0 / max(o[5], 1), o[14] * 256.0 / max(o[6], 1)))
    bt.update_horizon(refs[:, N + 1 + k].contiguous())
