"""Soak: warm control steps of whole batches on the GPU against the CPU port, bit for bit.
usage: python tests/soak_warm.py B N steps seed [B N steps seed ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import portdriver as P
from nmpc_amd import synthetic as S
from nmpc_amd.batch import Batch
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import read_by_solver, same_bits

PREC = os.environ.get("SOAK_PRECISION", "double")     # float: with NMPC_AMD_LIB=.../libcnmpc_f32.so
BOUNDS = os.environ.get("SOAK_TIGHT_BOUNDS")           # narrow control bounds: many active constraints
DUMPS = int(os.environ.get("SOAK_DUMPS", "0"))         # compare the final state dumps of this many instances
RTI = os.environ.get("SOAK_RTI")                       # real-time-iteration preparation before every solve
HOSTILE = float(os.environ.get("SOAK_HOSTILE", "0"))   # fraction of instances with non-finite / huge inputs
NOISE = float(os.environ.get("SOAK_NOISE", "1"))       # measurement deviations (position, velocity, rates) times this:
                                                       # large values drive solves into their failure paths
args = [int(a) for a in sys.argv[1:]]
for B, N, steps, seed in zip(args[0::4], args[1::4], args[2::4], args[3::4]):
    d = S.make_batch(B, N, extra_steps=steps, seed=seed)
    cfg = S.setup_args()
    if NOISE != 1:
        base = d["refs"][:, :steps + 1, :13]
        for sl in (slice(0, 6), slice(10, 13)):
            d["meas"][:, :, sl] = base[:, :, sl] + NOISE * (d["meas"][:, :, sl] - base[:, :, sl])
    if HOSTILE:
        # zero, huge, infinite and NaN entries in the measurements, the wind and the
        # reference points of some instances
        hr = np.random.default_rng(seed + 1)
        special = np.array([0.0, -0.0, 1e-300, 1e150, -1e200, np.inf, -np.inf, np.nan, 1e6, -1e4])
        pick = hr.random(B) < HOSTILE
        for key in ("meas", "wind", "refs"):
            arr = d[key]
            m = (hr.random(arr.shape) < 0.02) & pick.reshape((B,) + (1,) * (arr.ndim - 1))
            arr[m] = special[hr.integers(0, len(special), int(m.sum()))]
    if BOUNDS:
        cfg.update(upper_control_bound=[0.62, 0.55, 0.55], lower_control_bound=[0.4, 0.45, 0.45])
    bt = Batch(B, N); bt.setup(**cfg); bt.set_rti(bool(RTI)); bt.initialise_horizon(); bt.set_wind_velocity(d["wind"])
    bt.set_reference(d["refs"][:, :N + 1])
    got = []
    for k in range(steps):
        bt.prepare(); bt.solve(d["meas"][:, k]); got.append(bt.get_controls()); bt.update_horizon(d["refs"][:, N + 1 + k])
    dumps = [(bt.stages(b), bt.globals_(b)) for b in range(min(DUMPS, B))]
    bt.close()

    def one(b):
        p = P.Port(N, PREC)
        p.setup(cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"],
                cfg["upper_control_bound"], cfg["lower_control_bound"])
        p.init(); p.set_wind(d["wind"][b])
        for i in range(N + 1):
            p.set_reference(d["refs"][b, i], i)
        out = []
        for k in range(steps):
            if RTI:
                p.prepare_rti()
            p.feedback(d["meas"][b, k]); u, st = p.controls(); out.append((u, st, p.num_iter()))
            p.update_horizon(d["refs"][b, N + 1 + k])
        if b < len(dumps):
            out.append(int((~same_bits(dumps[b][0], p.stages())).sum() + (~same_bits(read_by_solver(dumps[b][1]), read_by_solver(p.globals_()))).sum()))
        return out
    with ThreadPoolExecutor(16) as pool:
        exp = list(pool.map(one, range(B)))
    bad = [(b, k) for b in range(B) for k in range(steps)
           if not (same_bits(got[k][0][b], exp[b][k][0]).all() and got[k][1][b] == exp[b][k][1]
                   and got[k][2][b] == exp[b][k][2])]
    if os.environ.get("SOAK_TOLERANT"):
        # a library whose arithmetic is not the reference's to the bit: solves whose status or
        # iteration count differs, and the largest relative deviation of the controls of the others
        differ = [(b, k) for b in range(B) for k in range(steps)
                  if got[k][1][b] != exp[b][k][1] or got[k][2][b] != exp[b][k][2]]
        skip = set(differ)
        rel = 0.0
        for b in range(B):
            for k in range(steps):
                if (b, k) not in skip and exp[b][k][1] == 0:
                    e = np.asarray(exp[b][k][0], dtype=np.float64)
                    rel = max(rel, float(np.max(np.abs(got[k][0][b] - e) / np.maximum(np.abs(e), 1e-300))))
        inst = len({b for b, _ in differ})
        print("tolerant: %d of %d solves (%d of %d instances) differ in status or iteration count; "
              "largest relative control deviation of the others %.3e; bitwise different controls: %d"
              % (len(differ), B * steps, inst, B, rel, len(bad)), flush=True)
    ok = sum(int((g[1] == 0).sum()) for g in got)
    dumped = [exp[b][steps] for b in range(len(dumps))]
    print(PREC, "rti" if RTI else "", "tight" if BOUNDS else "", "noise x%g" % NOISE if NOISE != 1 else "", "hostile %g" % HOSTILE if HOSTILE else "", "B %d N %d steps %d seed %d: %d mismatches, %d of %d solves converged" % (B, N, steps, seed, len(bad), ok, B * steps), bad[:5],
          "; state dumps of %d instances: %d entries differ" % (len(dumps), sum(dumped)) if dumps else "", flush=True)
