"""Golden-vector generator (runs only where the reference tree is present).

Drives the reference's own pure-C controller, compiled in fp64 by
`make -C oracle ref N=<horizon>` into oracle/_ref/, on seeded inputs and stores
inputs plus the reference's outputs as compressed .npz fixtures next to this
file.  The fixtures are data only; nothing of the reference's source travels.
The *_f32 fixtures come from the reference in its shipped single-precision
configuration (`make -C oracle ref PREC=float`); their arrays are float32.

    python tests/golden/generate.py [part ...]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from refdriver import Ref  # noqa: E402
from nmpc_amd import synthetic as S  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def gen_model(precision="double", name="model"):
    rng = np.random.default_rng(20131118)
    r = Ref(100, precision)
    n = 2048
    x, u, w = (a.astype(r.dtype) for a in S.random_states(n, rng))
    acc = np.stack([r.x8_dynamics(x[i], u[i], w[i]) for i in range(n)])
    xdot = np.stack([r.state_model(x[i], u[i], w[i]) for i in range(n)])
    xplus = np.stack([r.rk4(x[i], u[i], w[i]) for i in range(n)])
    # state difference, including pairs whose error quaternion has w < 0
    s1 = x.copy()
    s2 = np.roll(x, 7, axis=0).copy()
    s2[::3, 6:10] = -s1[::3, 6:10] + rng.normal(0, 0.05, (len(s1[::3]), 4)).astype(r.dtype)
    delta = np.stack([r.state_to_delta(s1[i], s2[i]) for i in range(n)])
    save(name, x=x, u=u, wind=w, accel=acc, xdot=xdot, xplus=xplus,
         s1=s1, s2=s2, delta=delta)


def gen_linearise(precision="double", name="linearise"):
    rng = np.random.default_rng(42)
    r = Ref(100, precision)
    n = 384
    x, u, w = (a.astype(r.dtype) for a in S.random_states(n, rng))
    xn = np.stack([r.rk4(x[i], u[i], w[i]) for i in range(n)])
    xn += rng.normal(0, 1e-3, xn.shape).astype(r.dtype)
    xn[::5] = np.roll(x, 3, axis=0)[::5]
    lower = np.array([0.0, -0.1, 0.05], dtype=r.dtype)
    upper = np.array([1.0, 0.9, 1.3], dtype=r.dtype)
    jac = np.zeros((n, 180), dtype=r.dtype)
    res = np.zeros((n, 12), dtype=r.dtype)
    for i in range(n):
        jac[i], res[i] = r.linearise(x[i], u[i], xn[i], w[i], lower, upper)
    save(name, x=x, u=u, xnext=xn, wind=w, lower=lower, upper=upper,
         jac=jac, res=res)


def run_sequence(name, horizon, refs, meas, wind_schedule, steps, weights=None,
                 bounds=None, dump=False, precision="double"):
    """wind_schedule: list of (first_step, wind3).  The wind set before loading
    the horizon is wind_schedule[0]."""
    cfg = S.setup_args()
    if weights:
        cfg.update(weights)
    if bounds:
        cfg.update(bounds)
    r = Ref(horizon, precision)
    dt = r.dtype
    refs, meas = np.asarray(refs, dtype=dt), np.asarray(meas, dtype=dt)
    r.setup(cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"],
            cfg["upper_control_bound"], cfg["lower_control_bound"])
    r.init()
    winds = np.zeros((steps, 3), dtype=dt)
    cur = np.asarray(wind_schedule[0][1], dtype=dt)
    sched = {int(s): np.asarray(w, dtype=dt) for s, w in wind_schedule}
    r.set_wind(cur)
    for i in range(horizon + 1):
        r.set_reference(refs[i], i)
    loaded = r.stages()[:, :192].copy() if dump else None     # C and c after loading
    u = np.zeros((steps, 3), dtype=dt)
    status = np.zeros(steps, dtype=np.int32)
    iters = np.zeros(steps, dtype=np.int32)
    ls = np.zeros(steps, dtype=np.int32)
    alpha = np.zeros(steps, dtype=dt)
    z0 = np.zeros((steps, 15), dtype=dt)
    lam_norm = np.zeros(steps)
    for k in range(steps):
        if k in sched:
            cur = sched[k]
        winds[k] = cur
        r.lib.nmpc_preparation_step()
        r.set_wind(cur)
        r.feedback(meas[k])
        u[k], status[k] = r.controls()
        iters[k] = r.num_iter()
        ls[k] = r.lib.ref_line_search_iter()
        alpha[k] = r.lib.ref_alpha()
        z0[k] = r.stages()[0, 238:253]
        lam_norm[k] = np.linalg.norm(r.gfield(r.globals_(), "lambda"))
        r.update_horizon(refs[horizon + 1 + k])
    out = dict(horizon=np.int32(horizon), refs=np.asarray(refs[:horizon + 1 + steps]),
               meas=np.asarray(meas[:steps]), winds=winds,
               state_weights=np.asarray(cfg["state_weights"], dtype=dt),
               control_weights=np.asarray(cfg["control_weights"], dtype=dt),
               terminal_weights=np.asarray(cfg["terminal_weights"], dtype=dt),
               upper=np.asarray(cfg["upper_control_bound"], dtype=dt),
               lower=np.asarray(cfg["lower_control_bound"], dtype=dt),
               u=u, status=status, iters=iters, ls_iters=ls, alpha=alpha, z0=z0,
               lam_norm=lam_norm)
    if dump:
        out["loaded_Cc"] = loaded
        out["final_stages"] = r.stages()
        out["final_globals"] = r.globals_()
    save(name, **out)
    print("    status", status.tolist(), "iters", iters.tolist(), "ls", ls.tolist())


def gen_sequences():
    N = 100
    # the single-instance check scenario: level flight, lateral + sink offset
    steps = 8
    refs = S.straight_reference(N + 1 + steps)
    meas = np.stack([S.straight_measurement(refs[k]) for k in range(steps)])
    run_sequence("seq_straight", N, refs, meas, [(0, [0, 0, 0])], steps, dump=True)

    # turning trajectories with measurement noise and wind changes mid-run
    steps = 50
    d = S.make_batch(6, N, extra_steps=steps, seed=5, turn_start=(0.3, 3.0))
    for b in range(6):
        w = d["wind"][b]
        sched = [(0, w), (10, 0.5 * w), (25, [w[1], -w[0], 0.3 * w[2]])]
        run_sequence("seq_turn%d" % b, N, d["refs"][b], d["meas"][b], sched, steps)

    # active control bounds
    d = S.make_batch(2, N, extra_steps=20, seed=9, turn_start=(0.3, 3.0))
    run_sequence("seq_bound", N, d["refs"][1], d["meas"][1], [(0, d["wind"][1])], 20,
                 bounds=dict(upper_control_bound=[0.62, 0.55, 0.55],
                             lower_control_bound=[0.4, 0.45, 0.45]))
    # identity weights: the solver's IDENTITY sparsity path
    run_sequence("seq_identity", N, d["refs"][1], d["meas"][1], [(0, d["wind"][1])], 8,
                 weights=dict(state_weights=[1] * 12, control_weights=[1] * 3,
                              terminal_weights=[1] * 12))
    # huge weights: Newton Hessian below the pivot tolerance -> regularisation
    run_sequence("seq_regularised", N, d["refs"][0], d["meas"][0], [(0, d["wind"][0])], 6,
                 weights=dict(state_weights=[1e16] * 12, control_weights=[1e16] * 3,
                              terminal_weights=[1e16] * 12), dump=True)
    # line search running to its iteration limit
    run_sequence("seq_linesearch", N, d["refs"][0], d["meas"][0], [(0, d["wind"][0])], 6,
                 weights=dict(state_weights=[1e10] * 12, control_weights=[1e10] * 3,
                              terminal_weights=[1e10] * 12))
    # measurement exactly on the reference
    refs = S.straight_reference(N + 1 + 5)
    run_sequence("seq_exact", N, refs, refs[:5, :13], [(0, [1.0, -2.0, 0.5])], 5)


def gen_indefinite():
    """Negative state weights: the Newton Hessian is not positive definite even
    after the solver's one regularisation, so the factorisation fails twice and
    the solve ends with a division-by-zero status (dual_qp.c:643-689)."""
    N = 100
    d = S.make_batch(1, N, extra_steps=3, seed=5)
    run_sequence("seq_indefinite", N, d["refs"][0], d["meas"][0], [(0, d["wind"][0])], 3,
                 weights=dict(state_weights=[1, 1, 1, -1, -1, -1, 1, 1, 1, 1, 1, 1]), dump=True)


def gen_reinit():
    """A second nmpc_init in the middle of a run (ccs-c66x/cnmpc.c:872-948): it
    rebuilds the stage QPs and the interval order from the weights and bounds of
    that moment, but the multipliers, the stage vectors' history and the Newton
    system are file statics that it does not clear (cnmpc.c:298, 801-870), so the
    first solve after it starts from what the run before left.  Phase 1: the
    default configuration, four control steps.  Phase 2: other weights and bounds,
    init, another wind, another trajectory, four control steps."""
    N, steps = 100, 4
    d = S.make_batch(2, N, extra_steps=steps, seed=11, turn_start=(0.3, 3.0))
    cfg1 = S.setup_args()
    cfg2 = dict(state_weights=[2, 2, 2, 1, 1, 1, 3, 3, 12, 0.5, 0.5, 8], control_weights=[0.2, 8e2, 9e2],
                terminal_weights=[1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1],
                upper_control_bound=[0.9, 0.95, 0.9], lower_control_bound=[0.1, 0.05, 0.1])
    r = Ref(N)
    out = dict(horizon=np.int32(N), steps=np.int32(steps), refs=d["refs"], meas=d["meas"][:, :steps],
               winds=d["wind"])
    for ph, cfg in enumerate((cfg1, cfg2)):
        for k, v in cfg.items():
            out["%s_%d" % (k, ph)] = np.asarray(v, dtype=np.float64)
    u = np.zeros((2, steps, 3))
    status = np.zeros((2, steps), dtype=np.int32)
    iters = np.zeros((2, steps), dtype=np.int32)
    for ph, cfg in enumerate((cfg1, cfg2)):
        r.setup(cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"],
                cfg["upper_control_bound"], cfg["lower_control_bound"])
        r.init()
        r.set_wind(d["wind"][ph])
        for i in range(N + 1):
            r.set_reference(d["refs"][ph, i], i)
        out["loaded_%d" % ph] = r.stages()
        for k in range(steps):
            r.feedback(d["meas"][ph, k])
            u[ph, k], status[ph, k] = r.controls()
            iters[ph, k] = r.num_iter()
            r.update_horizon(d["refs"][ph, N + 1 + k])
        out["final_stages_%d" % ph] = r.stages()
        out["final_globals_%d" % ph] = r.globals_()
    save("seq_reinit", u=u, status=status, iters=iters, **out)
    print("    status", status.tolist(), "iters", iters.tolist())


def gen_single_precision():
    """The reference in its shipped single-precision configuration."""
    gen_model("float", "model_f32")
    gen_linearise("float", "linearise_f32")
    N = 100
    steps = 8
    refs = S.straight_reference(N + 1 + steps)
    meas = np.stack([S.straight_measurement(refs[k]) for k in range(steps)])
    run_sequence("seq_straight_f32", N, refs, meas, [(0, [0, 0, 0])], steps, dump=True,
                 precision="float")
    steps = 30
    d = S.make_batch(4, N, extra_steps=steps, seed=5, turn_start=(0.3, 3.0))
    for b in range(4):
        w = d["wind"][b]
        sched = [(0, w), (10, 0.5 * w), (20, [w[1], -w[0], 0.3 * w[2]])]
        run_sequence("seq_turn%d_f32" % b, N, d["refs"][b], d["meas"][b], sched, steps,
                     dump=(b == 0), precision="float")
    run_sequence("seq_bound_f32", N, d["refs"][1], d["meas"][1], [(0, d["wind"][1])], 12,
                 bounds=dict(upper_control_bound=[0.62, 0.55, 0.55],
                             lower_control_bound=[0.4, 0.45, 0.45]), dump=True, precision="float")
    run_sequence("seq_identity_f32", N, d["refs"][2], d["meas"][2], [(0, d["wind"][2])], 8,
                 weights=dict(state_weights=[1] * 12, control_weights=[1] * 3,
                              terminal_weights=[1] * 12), precision="float")


def gen_long_horizon():
    from refdriver import have_ref
    if not have_ref(500):
        print("libref_n500.so missing: run `make -C oracle ref N=500`")
        return
    N = 500
    d = S.make_batch(2, N, extra_steps=3, seed=3, turn_start=(0.3, 9.0))
    for b in range(2):
        run_sequence("seq_n500_%d" % b, N, d["refs"][b], d["meas"][b],
                     [(0, d["wind"][b])], 3)


if __name__ == "__main__":
    import sys
    parts = dict(model=gen_model, linearise=gen_linearise, sequences=gen_sequences,
                 indefinite=gen_indefinite, reinit=gen_reinit, long_horizon=gen_long_horizon,
                 single_precision=gen_single_precision)
    for name in (sys.argv[1:] or list(parts)):      # no argument: everything
        parts[name]()
