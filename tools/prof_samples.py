"""Profiling workload: the stage linearisation of 2^20 independent samples
(device-resident inputs and outputs), a few launches; used under rocprofv3."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nmpc_amd import _lib
from nmpc_amd import synthetic as S

n = int(os.environ.get("PROF_SAMPLES", str(1 << 20)))
steps = int(os.environ.get("PROF_STEPS", "3"))
dev = torch.device("cuda:0")
x, u, w = S.random_states(n, np.random.default_rng(S.DEFAULT_SEED))
xn = np.roll(x, 1, axis=0)
tx, tu, tn, tw = (torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (x, u, xn, w))
jac = torch.empty((n, 180), dtype=torch.float64, device=dev)
res = torch.empty((n, 12), dtype=torch.float64, device=dev)
lib = _lib.load()
lo, up = np.zeros(3), np.ones(3)
p = lambda t: C.c_void_p(t.data_ptr())
for _ in range(steps):
    _lib.check(lib, lib.nmpc_linearise_samples(0, None, n, p(tx), p(tu), p(tn), p(tw),
                                               lo.ctypes.data_as(_lib.PREAL),
                                               up.ctypes.data_as(_lib.PREAL), p(jac), p(res)))
torch.cuda.synchronize()
print("done", float(res.sum()))
