"""GPU parity tests (pytest -m gpu): libcnmpc.so, called through its C ABI,
against the committed golden vectors of the compiled reference and against
the CPU oracle port on the same seeded inputs.

Bar: controls and Jacobians within 1e-9 relative of the reference fp64 CPU
path and identical QP iteration counts.  The kernels are built without FMA
contraction and reproduce the reference's operation order, so these tests
assert the stronger property -- bit-for-bit equality -- for every quantity.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import SEQUENCES, assert_same, assert_same_state, same_bits, config_of, drive_gpu_batch, golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rk4_samples_golden():
    from nmpc_amd.batch import rk4_samples
    g = golden("model")
    out = rk4_samples(g["x"], g["u"], g["wind"])
    assert_same(out, g["xplus"], "x+ (RK4)")


def test_linearise_samples_golden():
    from nmpc_amd.batch import linearise_samples
    g = golden("linearise")
    jac, res = linearise_samples(g["x"], g["u"], g["xnext"], g["wind"], g["lower"], g["upper"])
    np.testing.assert_allclose(jac, g["jac"], rtol=1e-9, atol=0)    # the stated bar
    assert_same(jac, g["jac"], "Jacobian")
    assert_same(res, g["res"], "residual")


def test_linearise_ragged_and_empty():
    """Sample counts that do not fill a 16-lane group / a workgroup, and n = 0."""
    from nmpc_amd import _lib
    from nmpc_amd.batch import linearise_samples
    g = golden("linearise")
    for n in (1, 3, 17, 255):
        jac, res = linearise_samples(g["x"][:n], g["u"][:n], g["xnext"][:n], g["wind"][:n],
                                     g["lower"], g["upper"])
        assert_same(jac, g["jac"][:n], "Jacobian n=%d" % n)
        assert_same(res, g["res"][:n], "residual n=%d" % n)
    lib = _lib.load()
    z = np.zeros(1)
    p = ctypes.c_void_p(z.ctypes.data)
    lo = g["lower"].ctypes.data_as(_lib.PREAL)
    assert lib.nmpc_linearise_samples(0, None, 0, p, p, p, p, lo, lo, p, p) == 0


@pytest.mark.parametrize("name", [s for s in SEQUENCES if not s.startswith("seq_turn")])
def test_sequence_golden(name):
    g = golden(name)
    dump = "final_stages" in g
    out = drive_gpu_batch([g], dump=dump)
    assert_same(out["status"][0], g["status"], name + " status")
    assert_same(out["iters"][0], g["iters"], name + " QP iterations")
    np.testing.assert_allclose(out["u"][0], g["u"], rtol=1e-9, atol=0)
    assert_same(out["u"][0], g["u"], name + " controls")
    assert_same(out["z0"][0], g["z0"], name + " z0")
    if dump:
        assert_same(out["loaded_Cc"][0], g["loaded_Cc"], name + " C,c after loading")
        assert_same(out["final_stages"][0], g["final_stages"], name + " final stage records")
        assert_same_state(out["final_globals"][0], g["final_globals"], name + " final solver state")


def test_turn_sequences_as_one_batch():
    """Six 50-step closed-loop sequences with wind changes run in lockstep as
    one batch: per-instance results must not depend on batch neighbours."""
    gs = [golden("seq_turn%d" % b) for b in range(6)]
    out = drive_gpu_batch(gs)
    for b, g in enumerate(gs):
        assert_same(out["status"][b], g["status"], "turn%d status" % b)
        assert_same(out["iters"][b], g["iters"], "turn%d QP iterations" % b)
        assert_same(out["u"][b], g["u"], "turn%d controls" % b)
        assert_same(out["z0"][b], g["z0"], "turn%d z0" % b)


def test_reference_interface_in_fresh_process():
    """The 24 reference symbols through the Python module mirror, in a child
    process so that the hidden instance starts from zeroed state."""
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from helpers import golden, same_bits
import nmpc_amd as nmpc
g = golden("seq_straight")
nmpc.init()
assert nmpc.HORIZON_LENGTH == 100 and nmpc.STEP_LENGTH == 0.02
nmpc.setup(state_weights=list(g["state_weights"]), control_weights=list(g["control_weights"]),
           terminal_weights=list(g["terminal_weights"]), upper_control_bound=list(g["upper"]),
           lower_control_bound=list(g["lower"]))
nmpc.initialise_horizon()
for i in range(nmpc.HORIZON_LENGTH + 1):
    nmpc.set_reference(list(g["refs"][i]), i)
for k in range(len(g["u"])):
    nmpc.prepare()
    nmpc.set_wind_velocity(list(g["winds"][k]))
    nmpc.solve(list(g["meas"][k]))
    u, st = nmpc.get_result()
    assert nmpc.get_controls() == u
    assert st == g["status"][k], (k, st)
    assert same_bits(np.array(u), g["u"][k]).all(), (k, u, g["u"][k])
    nmpc.update_horizon(list(g["refs"][101 + k]))
# plant integration (c/cnmpc.cpp:173-180) against the golden RK4 vectors
m = golden("model")
for i in range(5):
    nmpc.set_wind_velocity(list(m["wind"][i]))
    s = nmpc.state
    s.position[:] = list(m["x"][i, 0:3]); s.velocity[:] = list(m["x"][i, 3:6])
    s.attitude[:] = list(m["x"][i, 6:10]); s.angular_velocity[:] = list(m["x"][i, 10:13])
    nmpc.integrate(0.02, list(m["u"][i]))
    got = np.array(list(s.position) + list(s.velocity) + list(s.attitude) + list(s.angular_velocity))
    # dt crosses the ABI as a C float (c/cnmpc.h:84): 0.02f != 1/50, so compare
    # against an RK4 step of exactly that length
    from nmpc_amd.batch import rk4_samples
    exp = rk4_samples(m["x"][i:i+1], m["u"][i:i+1], m["wind"][i:i+1], delta=float(np.float32(0.02)))[0]
    assert same_bits(got, exp).all(), (i, got, exp)
print("child ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout + r.stderr


def _port_cold(d, N, threads=16):
    import portdriver as P
    from nmpc_amd import synthetic as S
    cfg = S.setup_args()
    return P.batch_cold_solve(
        N, cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"],
        cfg["lower_control_bound"], cfg["upper_control_bound"], d["refs"][:, :N + 1],
        d["meas"][:, 0], d["wind"], threads=threads)


def _gpu_cold(d, N):
    from nmpc_amd import synthetic as S
    from nmpc_amd.batch import Batch
    B = d["refs"].shape[0]
    bt = Batch(B, N)
    bt.setup(**S.setup_args())
    bt.initialise_horizon()
    bt.set_wind_velocity(d["wind"])
    bt.set_reference(d["refs"][:, :N + 1])
    bt.solve(d["meas"][:, 0])
    out = bt.get_controls()
    return bt, out


def test_full_size_cold_solve_4096(port_built):
    """The benchmark configuration (4096 instances, horizon 100): every
    instance against the CPU oracle, plus a permutation property."""
    from nmpc_amd import synthetic as S
    N, B = 100, 4096
    d = S.make_batch(B, N, seed=S.DEFAULT_SEED)
    bt, (u, st, it) = _gpu_cold(d, N)
    eu, est, eit, _ = _port_cold(d, N)
    assert_same(st, est, "status")
    assert_same(it, eit, "QP iterations")
    assert_same(u, eu, "controls")
    assert (st == 0).sum() > B // 2
    bt.close()
    # reversed instance order gives the reversed result
    dr = {k: np.ascontiguousarray(v[::-1]) for k, v in d.items()}
    bt2, (u2, st2, it2) = _gpu_cold(dr, N)
    assert_same(u2[::-1], u, "controls under instance permutation")
    assert_same(it2[::-1], it, "iterations under instance permutation")
    bt2.close()


def test_large_batch_20000(port_built):
    """20 000 instances: more line searches are handed over per Newton
    iteration than the hand-over kernels have workgroups, and the Newton
    kernel runs several rounds of wavefronts per SIMD."""
    from nmpc_amd import synthetic as S
    N, B = 100, 20000
    d = S.make_batch(B, N, seed=77)
    bt, (u, st, it) = _gpu_cold(d, N)
    eu, est, eit, _ = _port_cold(d, N)
    assert_same(st, est, "status")
    assert_same(it, eit, "QP iterations")
    assert_same(u, eu, "controls")
    bt.close()


def test_long_horizon_batch(port_built):
    """Horizon 500 (stage-parallel QP work): 24 instances against the oracle."""
    from nmpc_amd import synthetic as S
    N, B = 500, 24
    d = S.make_batch(B, N, seed=8, turn_start=(0.3, 9.0))
    bt, (u, st, it) = _gpu_cold(d, N)
    eu, est, eit, _ = _port_cold(d, N)
    assert_same(st, est, "status")
    assert_same(it, eit, "QP iterations")
    assert_same(u, eu, "controls")
    bt.close()


def test_small_and_odd_shapes(port_built):
    """Shapes that leave wavefronts partly empty: 3 instances x 7 stages,
    1 instance x 2 stages."""
    import portdriver as P
    from nmpc_amd import synthetic as S
    # 106 / 120: largest horizon the line-search hand-over kernels take / first one past it
    # (2, 5, 12) and (3, 70, 75): more horizon shifts than stages, every ring offset wraps
    for B, N, steps in ((3, 7, 6), (1, 2, 4), (2, 3, 4), (5, 65, 3), (4, 106, 3), (4, 120, 3), (2, 5, 12),
                        (3, 70, 75)):
        d = S.make_batch(B, N, extra_steps=steps, seed=B * 100 + N, turn_start=(0.05, 0.5))
        gs = []
        cfg = S.setup_args()
        for b in range(B):
            gs.append(dict(horizon=np.int32(N), refs=d["refs"][b], meas=d["meas"][b],
                           winds=np.tile(d["wind"][b], (steps, 1)), u=np.zeros((steps, 3)),
                           state_weights=np.array(cfg["state_weights"], float),
                           control_weights=np.array(cfg["control_weights"], float),
                           terminal_weights=np.array(cfg["terminal_weights"], float),
                           upper=np.array(cfg["upper_control_bound"], float),
                           lower=np.array(cfg["lower_control_bound"], float)))
        out = drive_gpu_batch(gs, dump=True)
        from helpers import drive_port
        for b in range(B):
            exp = drive_port(gs[b], dump=True)
            tag = "B%d N%d inst %d " % (B, N, b)
            assert_same(out["status"][b], exp["status"], tag + "status")
            assert_same(out["iters"][b], exp["iters"], tag + "iterations")
            assert_same(out["u"][b], exp["u"], tag + "controls")
            assert_same(out["final_stages"][b], exp["final_stages"], tag + "stage records")
            assert_same_state(out["final_globals"][b], exp["final_globals"], tag + "solver state")


def test_model_and_linearisation_on_hostile_inputs(port_built):
    """Huge, tiny, zero, infinite and NaN entries in state, controls, wind and
    successor: the integration and the stage linearisation give what the port
    gives, entry for entry (a NaN for a NaN), also where the root-and-reciprocal
    shortcut of the aircraft model has to stand aside."""
    import portdriver as P
    from nmpc_amd import synthetic as S
    from nmpc_amd.batch import linearise_samples, rk4_samples
    n = 3000
    rng = np.random.default_rng(77)
    x, u, w = S.random_states(n, rng, spread=3.0)
    xn = np.roll(x, 1, axis=0).copy()
    special = np.array([0.0, -0.0, 1e-310, -1e-300, 1e150, -1e200, 1e308, np.inf, -np.inf, np.nan,
                        1.0, -1.0, 1e-20])
    hit = rng.random(n) < 0.5
    for arr in (x, u, w, xn):
        m = (rng.random(arr.shape) < 0.08) & hit[:, None]
        arr[m] = special[rng.integers(0, len(special), int(m.sum()))]
    lo, up = np.zeros(3), np.ones(3)
    xp = rk4_samples(x, u, w)
    jac, res = linearise_samples(x, u, xn, w, lo, up)
    assert (~np.isfinite(xp)).any(axis=1).sum() > n // 10           # the hostile part is there
    for i in range(n):
        assert_same(xp[i], np.asarray(P.rk4(x[i], u[i], w[i])), "x+ of sample %d" % i)
        ej, er = P.linearise(x[i], u[i], xn[i], w[i], lo, up)
        assert_same(jac[i].ravel(), np.asarray(ej).ravel(), "Jacobian of sample %d" % i)
        assert_same(res[i], np.asarray(er), "residual of sample %d" % i)


def test_million_sample_linearisation(port_built):
    """2^20 (x, u, wind) samples through the stage-linearisation kernel:
    a strided subset against the oracle and batch-position independence."""
    import portdriver as P
    from nmpc_amd import synthetic as S
    from nmpc_amd.batch import linearise_samples
    n = 1 << 20
    rng = np.random.default_rng(5)
    x, u, w = S.random_states(n, rng)
    xn = np.roll(x, 1, axis=0)
    lo, up = [0.0, 0.0, 0.0], [1.0, 1.0, 1.0]
    jac, res = linearise_samples(x, u, xn, w, lo, up)
    for i in range(0, n, 4099):
        ej, er = P.linearise(x[i], u[i], xn[i], w[i], lo, up)
        assert_same(jac[i], ej, "Jacobian sample %d" % i)
        assert_same(res[i], er, "residual sample %d" % i)
    perm = rng.permutation(n)[: 1 << 16]
    j2, r2 = linearise_samples(x[perm], u[perm], xn[perm], w[perm], lo, up)
    assert_same(j2, jac[perm], "Jacobian under permutation")
    assert_same(r2, res[perm], "residual under permutation")


def test_device_resident_inputs_match_host_inputs():
    """torch CUDA tensors are consumed in place and give the same results."""
    import torch
    from nmpc_amd import synthetic as S
    from nmpc_amd.batch import Batch
    N, B = 100, 64
    d = S.make_batch(B, N, extra_steps=2, seed=21)
    _, (u, st, it) = _gpu_cold(d, N)
    dev = torch.device("cuda:0")
    bt = Batch(B, N, stream=torch.cuda.current_stream().cuda_stream)
    bt.setup(**S.setup_args())
    bt.initialise_horizon()
    bt.set_wind_velocity(torch.from_numpy(d["wind"]).to(dev))
    bt.set_reference(torch.from_numpy(np.ascontiguousarray(d["refs"][:, :N + 1])).to(dev))
    bt.solve(torch.from_numpy(np.ascontiguousarray(d["meas"][:, 0])).to(dev))
    tu = torch.zeros((B, 3), dtype=torch.float64, device=dev)
    ts = torch.zeros(B, dtype=torch.int32, device=dev)
    ti = torch.zeros(B, dtype=torch.int32, device=dev)
    bt.get_controls_into(tu, ts, ti)
    assert_same(tu.cpu().numpy(), u, "controls")
    assert_same(ts.cpu().numpy(), st, "status")
    assert_same(ti.cpu().numpy(), it, "iterations")
    bt.close()


def test_piecewise_reference_loading_and_reset():
    """Reference points loaded in several calls, one by one, or all at once
    leave identical solver state; stage records right after init hold zero
    Jacobians; reset + init reproduces the cold solve."""
    from nmpc_amd import synthetic as S
    from nmpc_amd.batch import Batch
    N, B = 20, 9
    d = S.make_batch(B, N, seed=33, turn_start=(0.05, 0.3))
    refs, meas = d["refs"][:, :N + 1], d["meas"][:, 0]

    def run(loader, bt=None):
        bt = bt or Batch(B, N)
        bt.setup(**S.setup_args())
        bt.initialise_horizon()
        bt.set_wind_velocity(d["wind"])
        loader(bt)
        bt.solve(meas)
        out = bt.get_controls()
        return bt, out, bt.stages(4), bt.globals_(4)

    bt0, o0, s0, g0 = run(lambda bt: bt.set_reference(refs))
    for loader in (
        lambda bt: [bt.set_reference(refs[:, a:b], first=a) for a, b in ((0, 8), (8, 9), (9, N + 1))],
        lambda bt: [bt.set_reference(refs[:, i:i + 1], first=i) for i in range(N + 1)],
    ):
        bt1, o1, s1, g1 = run(loader)
        for a, b, what in zip(o0, o1, ("controls", "status", "iterations")):
            assert_same(b, a, what)
        assert_same(s1, s0, "stage records")
        assert_same_state(g1, g0, "solver state")
        bt1.close()
    # a fresh init leaves zero Jacobians whatever was there before
    bt0.initialise_horizon()
    st = bt0.stages(2)
    assert not st[:, :192].any()
    bt0.set_reference(refs[:, :5])
    st = bt0.stages(2)
    assert st[:4, :192].any(axis=1).all() and not st[4:, :192].any()
    # reset returns to the state after create: everything reads as zero, also
    # right after the next init (apart from what init itself sets)
    bt0.reset()
    assert not bt0.stages(3)[:, 180:].any()         # (Jacobians are cleared by init)
    bt0.reset()
    bt0.initialise_horizon()
    st = bt0.stages(3)
    assert not st[:, 238:283].any()                 # z, y
    assert (st[:-1, 192:204] == -1e12).all()        # state bounds set by init
    bt0.reset()
    _, o2, s2, g2 = run(lambda bt: bt.set_reference(refs), bt0)
    for a, b, what in zip(o0, o2, ("controls", "status", "iterations")):
        assert_same(b, a, what + " after reset")
    assert_same(s2, s0, "stage records after reset")
    bt0.close()


@pytest.mark.parametrize("copies,watch", [(1, 0), (5, 3)])
def test_second_initialisation_keeps_solver_state(copies, watch):
    """init -> control steps -> init with other weights and bounds -> control
    steps: the second nmpc_init rebuilds the stage QPs and the interval order
    but keeps multipliers, stage history and Newton system
    (ccs-c66x/cnmpc.c:298, 801-870).  Against the sequence recorded from the
    compiled reference, full state after loading and at the end of both phases
    (the library's ring offsets, postponed clears and postponed re-staging all
    cross the second init here)."""
    from helpers import BatchAsController, drive_reinit
    g = golden("seq_reinit")
    ctrl = BatchAsController(int(g["horizon"]), copies, watch)
    drive_reinit(ctrl, g)
    ctrl.close()


def test_weights_and_bounds_per_instance(port_built):
    """Instances of one batch with different weights and control bounds
    (nmpc_batch_set_instance_weights / _control_bounds; the reference keeps them
    per controller, ccs-c66x/cnmpc.c:988-1020): every instance against a port
    controller configured the same way and taken through the same calls --
    weights latched by the initialisation, bounds read by it, by every later
    linearisation and by every feedback step, a batch-wide setter writing every
    instance again.  Two weight sets and three bound sets; inputs from host and
    from device memory."""
    import torch
    import portdriver as P
    from nmpc_amd import synthetic as S
    from nmpc_amd.batch import Batch
    N, B, steps = 30, 7, 5
    d = S.make_batch(B, N, extra_steps=steps, seed=606, turn_start=(0.1, 0.5))
    base = S.setup_args()
    w2 = dict(state_weights=[2, 2, 2, 1, 1, 1, 3, 3, 12, 0.5, 0.5, 8], control_weights=[0.2, 8e2, 9e2],
              terminal_weights=[1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1])
    ident = dict(state_weights=[1] * 12, control_weights=[1] * 3, terminal_weights=[1] * 12)
    b2 = dict(upper_control_bound=[0.9, 0.95, 0.9], lower_control_bound=[0.1, 0.05, 0.1])
    b3 = dict(upper_control_bound=[0.62, 0.55, 0.55], lower_control_bound=[0.4, 0.45, 0.45])
    # configuration of every instance at the first initialisation
    cfg = [dict(base) for _ in range(B)]
    for i in (2, 3):
        cfg[i].update(w2); cfg[i].update(b2)
    cfg[4].update(b3)
    cfg[5].update(ident)
    cfg[6].update(w2)
    arr = lambda key, idx: np.array([cfg[i][key] for i in idx], dtype=np.float64)

    bt = Batch(B, N)
    bt.setup(**base)
    bt.setup_instances(2, state_weights=arr("state_weights", range(2, 7)),
                       control_weights=arr("control_weights", range(2, 7)),
                       terminal_weights=arr("terminal_weights", range(2, 7)))
    dev = torch.device("cuda", 0)
    bt.setup_instances(2, upper_control_bound=torch.from_numpy(arr("upper_control_bound", range(2, 5))).to(dev),
                       lower_control_bound=torch.from_numpy(arr("lower_control_bound", range(2, 5))).to(dev))
    ports = []
    for i in range(B):
        p = P.Port(N)
        c = cfg[i]
        p.setup(c["state_weights"], c["control_weights"], c["terminal_weights"],
                c["upper_control_bound"], c["lower_control_bound"])
        ports.append(p)

    def load_and_run(first_step, nsteps, refs0):
        bt.initialise_horizon()
        bt.set_wind_velocity(d["wind"])
        bt.set_reference(d["refs"][:, refs0:refs0 + N + 1])
        for i, p in enumerate(ports):
            p.init()
            p.set_wind(d["wind"][i])
            for j in range(N + 1):
                p.set_reference(d["refs"][i, refs0 + j], j)
        for i, p in enumerate(ports):
            assert_same(bt.stages(i), p.stages(), "stage records after loading, instance %d" % i)
        for k in range(first_step, first_step + nsteps):
            if k == first_step + 1:
                # bounds changed in the middle of the run: instance 1 alone, through the
                # per-instance setter; they reach the stage linearised next and the next feedback step
                bt.setup_instances(1, upper_control_bound=np.array([b3["upper_control_bound"]]),
                                   lower_control_bound=np.array([b3["lower_control_bound"]]))
                c = cfg[1]
                c.update(b3)
                ports[1].setup(c["state_weights"], c["control_weights"], c["terminal_weights"],
                               c["upper_control_bound"], c["lower_control_bound"])
            bt.solve(d["meas"][:, k])
            u, st, it = bt.get_controls()
            for i, p in enumerate(ports):
                p.feedback(d["meas"][i, k])
                eu, est = p.controls()
                what = "instance %d step %d" % (i, k)
                assert st[i] == est and it[i] == p.num_iter(), what
                assert_same(u[i], eu, "controls, " + what)
            bt.update_horizon(d["refs"][:, refs0 + N + 1 + k - first_step])
            for i, p in enumerate(ports):
                p.update_horizon(d["refs"][i, refs0 + N + 1 + k - first_step])
        for i, p in enumerate(ports):
            assert_same(bt.stages(i), p.stages(), "final stage records, instance %d" % i)
            assert_same_state(bt.globals_(i), p.globals_(), "final solver state, instance %d" % i)
        return u

    u1 = load_and_run(0, 3, 0)
    assert len({tuple(r) for r in u1}) == B            # the configurations do make a difference
    # a batch-wide setter writes every instance; weights wait for the next initialisation
    bt.setup(**dict(base, **w2))
    for c, p in zip(cfg, ports):
        c.update(base); c.update(w2)
        p.setup(c["state_weights"], c["control_weights"], c["terminal_weights"],
                c["upper_control_bound"], c["lower_control_bound"])
    load_and_run(3, 2, 0)
    bt.close()


def test_closed_loop_with_device_plant(port_built):
    """Controllers and simulated vehicles both on the GPU: feedback on the plant
    state, one RK4 step of the plant under the returned controls and a wind
    that differs from the estimate, horizon shift -- against the same loop
    built from the CPU oracle's controller and RK4."""
    import portdriver as P
    from nmpc_amd import synthetic as S
    from nmpc_amd.batch import Batch
    N, B, steps, dt = 30, 6, 8, 1.0 / 50.0
    d = S.make_batch(B, N, extra_steps=steps, seed=404, turn_start=(0.05, 0.4))
    rng = np.random.default_rng(9)
    true_wind = d["wind"] + rng.normal(0.0, 1.0, (B, 3))
    cfg = S.setup_args()

    bt = Batch(B, N)
    bt.setup(**cfg)
    bt.initialise_horizon()
    bt.set_wind_velocity(d["wind"])
    bt.set_reference(d["refs"][:, :N + 1])
    bt.plant_set_state(d["meas"][:, 0])
    gu, gx = [], []
    for k in range(steps):
        bt.prepare()
        bt.solve_from_plant()
        u, st, it = bt.get_controls()
        bt.plant_integrate(dt, wind=true_wind)
        bt.update_horizon(d["refs"][:, N + 1 + k])
        gu.append(u)
        gx.append(bt.plant_state())
    bt.close()

    for b in range(B):
        c = P.Port(N)
        c.setup(cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"],
                cfg["upper_control_bound"], cfg["lower_control_bound"])
        c.init()
        c.set_wind(d["wind"][b])
        for i in range(N + 1):
            c.set_reference(d["refs"][b, i], i)
        x = d["meas"][b, 0].copy()
        for k in range(steps):
            c.feedback(x)
            u, _ = c.controls()
            x = P.rk4(x, u, true_wind[b], dt)
            c.update_horizon(d["refs"][b, N + 1 + k])
            assert_same(gu[k][b], np.asarray(u), "controls, instance %d step %d" % (b, k))
            assert_same(gx[k][b], np.asarray(x), "plant state, instance %d step %d" % (b, k))


def test_relinearise_matches_resetting_every_reference_point(port_built):
    """nmpc_batch_relinearise after a wind change and two horizon shifts equals
    calling set_reference_point again for every stored point on the oracle."""
    import portdriver as P
    from nmpc_amd import synthetic as S
    from nmpc_amd.batch import Batch
    N, B, steps = 25, 5, 5
    d = S.make_batch(B, N, extra_steps=steps, seed=515, turn_start=(0.05, 0.3))
    cfg = S.setup_args()
    wind2 = d["wind"] + np.array([2.0, -1.5, 0.5])
    relin_at = 2

    bt = Batch(B, N)
    bt.setup(**cfg)
    bt.initialise_horizon()
    bt.set_wind_velocity(d["wind"])
    bt.set_reference(d["refs"][:, :N + 1])
    got = []
    for k in range(steps):
        if k == relin_at:
            bt.set_wind_velocity(wind2)
            bt.relinearise()
        bt.solve(d["meas"][:, k])
        got.append(bt.get_controls())
        bt.update_horizon(d["refs"][:, N + 1 + k])
    gs = [bt.stages(b) for b in range(B)]
    gg = [bt.globals_(b) for b in range(B)]
    bt.close()

    for b in range(B):
        c = P.Port(N)
        c.setup(cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"],
                cfg["upper_control_bound"], cfg["lower_control_bound"])
        c.init()
        c.set_wind(d["wind"][b])
        for i in range(N + 1):
            c.set_reference(d["refs"][b, i], i)
        for k in range(steps):
            if k == relin_at:
                c.set_wind(wind2[b])
                for i in range(N + 1):
                    c.set_reference(d["refs"][b, k + i], i)
            c.feedback(d["meas"][b, k])
            u, st = c.controls()
            assert_same(got[k][0][b], u, "controls, instance %d step %d" % (b, k))
            assert got[k][1][b] == st
            assert got[k][2][b] == c.num_iter()
            c.update_horizon(d["refs"][b, N + 1 + k])
        assert_same(gs[b], c.stages(), "stage records, instance %d" % b)
        assert_same_state(gg[b], c.globals_(), "solver state, instance %d" % b)


def _interpolate_like_the_harness(t, points):
    """interpolate_reference of scripts/nmpc-xplane-synthetic.py:62-78 restated for
    one sample time (plain Python floats): bisect on the waypoint times, linear
    blend of all 17 entries, attitude scaled by 1 / |q| (scripts/vectors.py:299-310)."""
    import bisect
    import math
    index = bisect.bisect([p[0] for p in points], t)
    if index >= len(points):
        return list(points[-1][1:])
    a, b = points[index - 1], points[index]
    delta = [y - x for x, y in zip(a, b)]
    r = (t - a[0]) / delta[0]
    new = [x + d * r for x, d in zip(a, delta)]
    mag = 0.0
    for j in range(7, 11):
        mag += new[j] * new[j]
    scale = 1.0 / math.sqrt(mag)
    for j in range(7, 11):
        new[j] = new[j] * scale
    return new[1:]


def test_reference_points_from_waypoints(port_built):
    """Reference points interpolated on the device equal the harness's host
    interpolation bit for bit, and so does everything computed from them."""
    from nmpc_amd import synthetic as S
    from nmpc_amd.batch import Batch
    N, B, dt = 40, 7, 1.0 / 50.0
    rng = np.random.default_rng(12)
    wp14 = S.waypoint_batch(B, rng, turn_start=(0.1, 0.6), turn_time=0.5)
    wp = np.concatenate([wp14, np.full((B, wp14.shape[1], 3), 0.5)], axis=-1)      # controls
    wp[:, -1, 0] = 0.9      # last leg ends inside the horizon on purpose
    t0 = 0.013
    host = np.array([[_interpolate_like_the_harness(t0 + i * dt, [list(p) for p in wp[b]])
                      for i in range(N + 1)] for b in range(B)])
    d = S.make_batch(B, N, seed=3)

    def run(load):
        bt = Batch(B, N)
        bt.setup(**S.setup_args())
        bt.initialise_horizon()
        bt.set_wind_velocity(d["wind"])
        load(bt)
        bt.solve(host[:, 0, :13] + 0.01)
        out = bt.get_controls()
        st, gl = bt.stages(B - 1), bt.globals_(B - 1)
        bt.close()
        return out, st, gl

    o_dev, s_dev, g_dev = run(lambda bt: bt.set_reference_from_waypoints(wp, t0, dt))
    o_two, s_two, g_two = run(lambda bt: [bt.set_reference_from_waypoints(wp, t0, dt, 0, 11),
                                          bt.set_reference_from_waypoints(wp, t0, dt, 11, N - 10)])
    o_host, s_host, g_host = run(lambda bt: bt.set_reference(host))
    for a, b, what in zip(o_host, o_dev, ("controls", "status", "iterations")):
        assert_same(b, a, what)
    assert_same(g_dev, g_host, "solver state incl. stored references")
    assert_same(s_dev, s_host, "stage records")
    assert_same(g_two, g_host, "solver state, waypoints loaded in two calls")


def test_plain_division_path_of_the_factorisation():
    """The Newton factorisation takes root, reciprocal and quotients of a column
    from one iteration and repeats a block with sqrt and plain divisions when a
    root's significand is all ones, which random data never produces.
    libcnmpc_repeat.so is the same source built to repeat every block: golden
    sequences (with their final factor dumps) and a seeded batch must come out
    bit for bit the same through it.  The same build makes the regrouped-sum
    step-length search give up before every odd trial, so that searches reach
    the row-ordered kernel half-way through, which well-scaled data rarely does."""
    lib = os.path.join(ROOT, "nmpc_amd", "libcnmpc_repeat.so")
    assert os.path.exists(lib), "build it with `make -C nmpc_amd/csrc` (target all)"
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
from helpers import assert_same, assert_same_state, drive_gpu_batch, golden
from nmpc_amd import _lib
assert _lib.LIB_PATH.endswith("libcnmpc_repeat.so")
for name in ("seq_straight", "seq_regularised", "seq_linesearch", "seq_bound", "seq_turn0", "seq_n500_0"):
    g = golden(name)
    dump = "final_stages" in g
    out = drive_gpu_batch([g], dump=dump)
    assert_same(out["status"][0], g["status"], name + " status")
    assert_same(out["iters"][0], g["iters"], name + " QP iterations")
    assert_same(out["u"][0], g["u"], name + " controls")
    if dump:
        assert_same_state(out["final_globals"][0], g["final_globals"], name + " final solver state")
import portdriver as P
from nmpc_amd import synthetic as S
from nmpc_amd.batch import Batch
N, B = 40, 64
d = S.make_batch(B, N, seed=5)
cfg = S.setup_args()
bt = Batch(B, N); bt.setup(**cfg); bt.initialise_horizon(); bt.set_wind_velocity(d["wind"])
bt.set_reference(d["refs"][:, :N + 1]); bt.solve(d["meas"][:, 0])
u, st, it = bt.get_controls()
eu, est, eit, _ = P.batch_cold_solve(N, cfg["state_weights"], cfg["control_weights"],
    cfg["terminal_weights"], cfg["lower_control_bound"], cfg["upper_control_bound"],
    d["refs"][:, :N + 1], d["meas"][:, 0], d["wind"], threads=8)
assert_same(st, est, "status"); assert_same(it, eit, "QP iterations"); assert_same(u, eu, "controls")
print("child ok")
""" % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"))
    env = dict(os.environ, NMPC_AMD_LIB=lib)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout + r.stderr


def test_two_batches_interleaved(port_built):
    """Two batches of different shapes in one process, their calls interleaved
    (each on its own stream): neither disturbs the other."""
    from nmpc_amd import synthetic as S
    from nmpc_amd.batch import Batch
    shapes = ((70, 100, 11), (33, 40, 12))
    data = [S.make_batch(B, N, extra_steps=2, seed=seed) for B, N, seed in shapes]
    bts = []
    for (B, N, _), d in zip(shapes, data):
        bt = Batch(B, N)
        bt.setup(**S.setup_args())
        bts.append(bt)
    for bt in bts:
        bt.initialise_horizon()
    for bt, d in zip(bts, data):
        bt.set_wind_velocity(d["wind"])
    for bt, (B, N, _), d in zip(bts, shapes, data):
        bt.set_reference(d["refs"][:, :N + 1])
    got = [[], []]
    for k in range(3):
        for bt, d in zip(bts, data):
            bt.solve(d["meas"][:, k])
        for i, (bt, (B, N, _), d) in enumerate(zip(bts, shapes, data)):
            got[i].append(bt.get_controls())
            if k < 2:
                bt.update_horizon(d["refs"][:, N + 1 + k])
    for i, ((B, N, _), d) in enumerate(zip(shapes, data)):
        alone = Batch(B, N)
        alone.setup(**S.setup_args())
        alone.initialise_horizon()
        alone.set_wind_velocity(d["wind"])
        alone.set_reference(d["refs"][:, :N + 1])
        for k in range(3):
            alone.solve(d["meas"][:, k])
            for a, b, what in zip(alone.get_controls(), got[i][k], ("controls", "status", "iterations")):
                assert_same(b, a, "batch %d step %d %s" % (i, k, what))
            if k < 2:
                alone.update_horizon(d["refs"][:, N + 1 + k])
        alone.close()
    eu, est, eit, _ = _port_cold(data[1], shapes[1][1])
    assert_same(got[1][0][0], eu, "first solve of batch 1 against the oracle")
    for bt in bts:
        bt.close()


def test_error_reporting():
    from nmpc_amd import _lib
    from nmpc_amd import synthetic as S
    from nmpc_amd.batch import Batch
    bt = Batch(2, 10)
    with pytest.raises(_lib.NmpcError):
        bt.solve(np.zeros((2, 13)))             # before initialise_horizon
    bt.setup(**S.setup_args())
    bt.initialise_horizon()
    with pytest.raises(_lib.NmpcError):
        bt.set_reference(np.zeros((2, 3, 16)), first=9)     # runs past the horizon
    with pytest.raises(ValueError):
        bt.set_wind_velocity(np.zeros((3, 3)))
    bt.close()
    h = ctypes.c_void_p()
    lib = _lib.load()
    assert lib.nmpc_batch_create(ctypes.byref(h), 1, 100, 99, None) == -1   # no such device


def test_kernel_timing_interface():
    from nmpc_amd import synthetic as S
    N, B = 100, 128
    d = S.make_batch(B, N, seed=2)
    from nmpc_amd.batch import Batch
    bt = Batch(B, N)
    bt.setup(**S.setup_args())
    bt.initialise_horizon()
    bt.set_wind_velocity(d["wind"])
    bt.set_profiling(True)
    bt.set_reference(d["refs"][:, :N + 1])
    bt.solve(d["meas"][:, 0])
    t = bt.kernel_times()
    assert t["linearise"][1] == 1 and t["linearise"][0] > 0
    assert t["newton_solve"][1] == 5 and t["qp_begin"][1] == 1
    u_all = bt.get_controls()
    # timing restricted to one kernel: the others report nothing, results unchanged
    bt.reset()
    bt.initialise_horizon()
    bt.set_wind_velocity(d["wind"])
    bt.set_profiling(True, kinds=["newton_solve"])
    bt.set_reference(d["refs"][:, :N + 1])
    bt.solve(d["meas"][:, 0])
    t = bt.kernel_times()
    assert t["newton_solve"][1] == 5 and t["newton_solve"][0] > 0
    assert all(n == 0 for k, (ms, n) in t.items() if k != "newton_solve")
    for a, b, what in zip(u_all, bt.get_controls(), ("controls", "status", "iterations")):
        assert_same(b, a, what)
    bt.set_profiling(False)
    bt.close()
