"""Host time of each call of a control step through the reference's interface (one controller)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmpc_amd as nmpc
from nmpc_amd import synthetic as S

N, W = 100, 300
d = S.make_batch(1, N, extra_steps=W + 2, seed=S.DEFAULT_SEED)
refs, meas, wind = d["refs"][0], d["meas"][0], d["wind"][0]
cfg = S.setup_args()
nmpc.init(); nmpc.setup(**{k: list(v) for k, v in cfg.items()})
nmpc.initialise_horizon(); nmpc.set_wind_velocity(list(wind))
for i in range(N + 1):
    nmpc.set_reference(list(refs[i]), i)
nmpc.prepare(); nmpc.solve(list(meas[0])); nmpc.get_controls()
names = ("update_horizon", "prepare", "set_wind_velocity", "solve", "get_result")
acc = {n: [] for n in names}
for k in range(W):
    r = list(refs[N + 1 + k]); m = list(meas[k + 1]); w = list(wind)
    t = [time.perf_counter()]
    nmpc.update_horizon(r); t.append(time.perf_counter())
    nmpc.prepare(); t.append(time.perf_counter())
    nmpc.set_wind_velocity(w); t.append(time.perf_counter())
    nmpc.solve(m); t.append(time.perf_counter())
    nmpc.get_result(); t.append(time.perf_counter())
    for i, n in enumerate(names):
        acc[n].append((t[i + 1] - t[i]) * 1e6)
for n in names:
    v = sorted(acc[n][50:])
    print("%-18s p50 %8.1f us" % (n, v[len(v) // 2]))
