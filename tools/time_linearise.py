"""Time of the stateless stage-linearisation kernel on as many samples as the
bench workload has stages (409 600), inputs resident on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ctypes as C
from nmpc_amd import _lib, synthetic as S

n = int(os.environ.get("N_SAMPLES", 409600))
rng = np.random.default_rng(1)
x, u, w = S.random_states(n, rng)
dev = torch.device("cuda:0")
tx = torch.from_numpy(x).to(dev); tu = torch.from_numpy(u).to(dev); tw = torch.from_numpy(w).to(dev)
tn = torch.roll(tx, 1, 0).contiguous()
jac = torch.zeros((n, 180), dtype=torch.float64, device=dev)
res = torch.zeros((n, 12), dtype=torch.float64, device=dev)
out = torch.zeros((n, 13), dtype=torch.float64, device=dev)
lib = _lib.load()
lo = (C.c_double * 3)(0, 0, 0); up = (C.c_double * 3)(1, 1, 1)
P = lambda t: C.c_void_p(t.data_ptr())
def lin():
    _lib.check(lib, lib.nmpc_linearise_samples(0, None, n, P(tx), P(tu), P(tn), P(tw), lo, up, P(jac), P(res)))
def rk4():
    _lib.check(lib, lib.nmpc_rk4_samples(0, None, n, P(tx), P(tu), P(tw), 0.02, P(out)))
for name, fn, flop in (("linearise", lin, 17000.0), ("rk4", rk4, 1000.0)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("%s: %.3f ms for %d samples, %.2f TFLOP/s" % (name, dt * 1e3, n, flop * n / dt / 1e12))
