"""Developer check (needs a GPU): libcnmpc.so against the CPU oracle port with
first-difference reporting.  Not part of the shipped path."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import portdriver as P  # noqa: E402
from refdriver import STAGE_FIELDS, global_fields  # noqa: E402
from nmpc_amd import synthetic as S  # noqa: E402
from nmpc_amd.batch import Batch, linearise_samples, rk4_samples  # noqa: E402


def report(name, a, b):
    same = a.shape == b.shape and a.tobytes() == b.tobytes()
    if not same:
        d = np.argwhere(a != b)
        rel = np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))
        print("  MISMATCH %s: %d of %d differ, first at %s gpu=%r cpu=%r maxrel=%.3e" % (
            name, len(d), a.size, d[0], a[tuple(d[0])], b[tuple(d[0])], rel))
    return same


def check_samples(n=4096):
    rng = np.random.default_rng(7)
    x, u, w = S.random_states(n, rng)
    out = rk4_samples(x, u, w)
    exp = np.stack([P.rk4(x[i], u[i], w[i]) for i in range(n)])
    ok = report("rk4", out, exp)
    xn = np.roll(x, 1, axis=0)
    lo, up = [0.0, -0.1, 0.05], [1.0, 0.9, 1.3]
    jac, res = linearise_samples(x, u, xn, w, lo, up)
    ej = np.zeros_like(jac); er = np.zeros_like(res)
    for i in range(n):
        ej[i], er[i] = P.linearise(x[i], u[i], xn[i], w[i], lo, up)
    ok &= report("jacobian", jac, ej)
    ok &= report("residual", res, er)
    print("samples:", "OK" if ok else "FAIL")
    return ok


def compare_state(bt, inst, p, tag):
    ok = True
    sa, sb = bt.stages(inst), p.stages()
    for name, (o, l) in STAGE_FIELDS.items():
        ok &= report("%s stage.%s" % (tag, name), sa[:, o:o + l], sb[:, o:o + l])
    from helpers import read_by_solver
    ga, gb = read_by_solver(bt.globals_(inst)), read_by_solver(p.globals_())
    gf, _ = global_fields(bt.N)
    for name, (o, l) in gf.items():
        ok &= report("%s global.%s" % (tag, name), ga[o:o + l], gb[o:o + l])
    return ok


def check_sequence(B=8, N=100, steps=6, seed=5, full=True, turn=(0.3, 3.0)):
    d = S.make_batch(B, N, extra_steps=steps, seed=seed, turn_start=turn)
    refs, meas, wind = d["refs"], d["meas"], d["wind"]
    bt = Batch(B, N)
    bt.setup(**S.setup_args())
    bt.initialise_horizon()
    bt.set_wind_velocity(wind)
    bt.set_reference(refs[:, :N + 1])
    ports = []
    for b in range(B):
        p = P.Port(N)
        a = S.setup_args()
        p.setup(a["state_weights"], a["control_weights"], a["terminal_weights"],
                a["upper_control_bound"], a["lower_control_bound"])
        p.init()
        p.set_wind(wind[b])
        for i in range(N + 1):
            p.set_reference(refs[b, i], i)
        ports.append(p)
    ok = True
    if full:
        for b in range(min(B, 2)):
            ok &= compare_state(bt, b, ports[b], "load b%d" % b)
    for k in range(steps):
        bt.solve(meas[:, k])
        u, st, it = bt.get_controls()
        for b in range(B):
            ports[b].feedback(meas[b, k])
            eu, est = ports[b].controls()
            same = np.asarray(u[b]).tobytes() == np.asarray(eu, np.asarray(u[b]).dtype).tobytes() and st[b] == est and it[b] == ports[b].num_iter()
            if not same:
                print("step %d inst %d: gpu u=%s st=%d it=%d | cpu u=%s st=%d it=%d info=%s" % (
                    k, b, u[b], st[b], it[b], eu, est, ports[b].num_iter(), bt.solver_info(b)))
                ok = False
        if full:
            for b in range(min(B, 2)):
                ok &= compare_state(bt, b, ports[b], "step %d b%d" % (k, b))
        if not ok:
            break
        bt.update_horizon(refs[:, N + 1 + k])
        for b in range(B):
            ports[b].update_horizon(refs[b, N + 1 + k])
    print("sequence B=%d N=%d steps=%d:" % (B, N, steps), "OK" if ok else "FAIL",
          "iters", it, "status", st)
    return ok


if __name__ == "__main__":
    t0 = time.time()
    ok = check_samples()
    ok &= check_sequence()
    print("ALL OK" if ok else "FAILED", "%.1fs" % (time.time() - t0))
    sys.exit(0 if ok else 1)
