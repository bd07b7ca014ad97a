"""Cycles of the two sweeps of the Newton solve, per wavefront, from the measurement build
(make -C nmpc_amd/csrc OBJDIR=../.build_stats DEF_f64=-DNMPC_SOLVE_STATS LIB_f64=../libcnmpc_stats.so ../libcnmpc_stats.so;
run with NMPC_AMD_LIB=nmpc_amd/libcnmpc_stats.so)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nmpc_amd import _lib
from nmpc_amd import synthetic as S
from nmpc_amd.batch import Batch

B = int(os.environ.get("PROF_B", "4096")); N = int(os.environ.get("PROF_N", "100"))
lib = _lib.load()
dev = torch.device("cuda:0")
d = S.make_batch(B, N, seed=S.DEFAULT_SEED)
bt = Batch(B, N, stream=torch.cuda.current_stream().cuda_stream)
bt.setup(**S.setup_args())
out = (C.c_ulonglong * 16)()
for rep in range(3):
    bt.reset(); bt.initialise_horizon(); bt.set_wind_velocity(d["wind"]); bt.set_reference(d["refs"][:, :N + 1])
    torch.cuda.synchronize()
    lib.nmpc_debug_solve_stats(out, 1)
    lib.nmpc_debug_team_stats(out, 1)
    bt.solve(d["meas"][:, 0]); bt.get_controls()
    torch.cuda.synchronize()
    lib.nmpc_debug_team_stats(out, 1)
    for role, name in enumerate(("factor", "substitution")):
        n = out[role * 4 + 3]
        if n:
            print("team wave %d %-12s sweeps %5d  cycles %9.0f  of them waiting %9.0f  looks %8.0f"
                  % (role, name, n, out[role * 4] / n, out[role * 4 + 1] / n, out[role * 4 + 2] / n))
    lib.nmpc_debug_solve_stats(out, 1)
    for name, o in (("first iteration + long parts", 0), ("short parts", 8)):
        n = out[o]
        if n:
            print("%-30s wavefronts %6d  bottom-up %8.0f cycles (max %8d)  top-down %8.0f (max %8d)  blocks factorised %6.1f"
                  % (name, n, out[o + 1] / n, out[o + 4], out[o + 2] / n, out[o + 5], out[o + 3] / n))
