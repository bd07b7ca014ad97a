"""GPU parity tests (pytest -m gpu) of shapes and call orders beyond
test_gpu_parity.py: the per-GPU share of the 8-GPU configuration and control
bounds changed between initialisation and reference loading."""
import numpy as np
import pytest

from helpers import assert_same, assert_same_state, same_bits

pytestmark = pytest.mark.gpu


def test_per_gpu_share_of_the_8_gpu_configuration(port_built):
    """32 768 instances, horizon 100: what each rank of the 262 144-instance
    configuration runs.  Every instance against the CPU oracle."""
    import portdriver as P
    from nmpc_amd import synthetic as S
    from nmpc_amd.batch import Batch
    N, B = 100, 32768
    d = S.make_batch(B, N, seed=S.DEFAULT_SEED + 3)
    cfg = S.setup_args()
    bt = Batch(B, N)
    bt.setup(**cfg)
    bt.initialise_horizon()
    bt.set_wind_velocity(d["wind"])
    bt.set_reference(d["refs"][:, :N + 1])
    bt.solve(d["meas"][:, 0])
    u, st, it = bt.get_controls()
    bt.close()
    eu, est, eit, _ = P.batch_cold_solve(
        N, cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"],
        cfg["lower_control_bound"], cfg["upper_control_bound"], d["refs"][:, :N + 1],
        d["meas"][:, 0], d["wind"], threads=16)
    assert_same(st, est, "status")
    assert_same(it, eit, "QP iterations")
    assert_same(u, eu, "controls")
    assert (st == 0).sum() > B // 2


def test_bounds_changed_between_init_and_partial_reference_loading(port_built):
    """nmpc_init fixes the stage bounds from the control bounds of that moment
    (ccs-c66x/cnmpc.c:920-931); bounds set afterwards reach only the stages
    linearised afterwards (cnmpc.c:1053-1062).  The library postpones part of
    its initialisation, which must not let later bounds leak into it."""
    import portdriver as P
    from nmpc_amd import synthetic as S
    from nmpc_amd.batch import Batch
    N, B = 12, 3
    d = S.make_batch(B, N, seed=91, turn_start=(0.05, 0.2))
    cfg = S.setup_args()
    lo2, up2 = [0.1, 0.2, 0.3], [0.9, 0.8, 0.7]
    bt = Batch(B, N)
    bt.setup(**cfg)
    bt.initialise_horizon()
    bt.set_wind_velocity(d["wind"])
    bt.setup(**dict(cfg, lower_control_bound=lo2, upper_control_bound=up2))
    bt.set_reference(d["refs"][:, :5])          # stages 0..3 only
    got = [bt.stages(b) for b in range(B)]
    bt.close()
    for b in range(B):
        c = P.Port(N)
        c.setup(cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"],
                cfg["upper_control_bound"], cfg["lower_control_bound"])
        c.init()
        c.set_wind(d["wind"][b])
        c.setup(cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"], up2, lo2)
        for i in range(5):
            c.set_reference(d["refs"][b, i], i)
        exp = c.stages()
        assert_same(got[b], exp, "stage records, instance %d" % b)
        assert (exp[6, 204:207] == 0.0).all() and (exp[1, 204:207] != 0.0).all()


def batch_against_port(B, N, steps, seed, dumped, mutate=None):
    """Control loop (feedback, controls, horizon shift) of a whole batch against the
    port, instance by instance: controls, status and iteration count of every
    instance and step; the complete stage records and solver state of the first
    `dumped` instances after the last step.  Returns the number of converged solves."""
    from concurrent.futures import ThreadPoolExecutor
    import portdriver as P
    from nmpc_amd import synthetic as S
    from nmpc_amd.batch import Batch
    d = S.make_batch(B, N, extra_steps=steps, seed=seed)
    if mutate:
        mutate(d)
    cfg = S.setup_args()
    bt = Batch(B, N)
    bt.setup(**cfg)
    bt.initialise_horizon()
    bt.set_wind_velocity(d["wind"])
    bt.set_reference(d["refs"][:, :N + 1])
    got = []
    for k in range(steps):
        bt.solve(d["meas"][:, k])
        got.append(bt.get_controls())
        bt.update_horizon(d["refs"][:, N + 1 + k])
    dumps = [(bt.stages(b), bt.globals_(b)) for b in range(dumped)]
    bt.close()

    def one(b):
        p = P.Port(N)
        p.setup(cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"],
                cfg["upper_control_bound"], cfg["lower_control_bound"])
        p.init()
        p.set_wind(d["wind"][b])
        for i in range(N + 1):
            p.set_reference(d["refs"][b, i], i)
        out = []
        for k in range(steps):
            p.feedback(d["meas"][b, k])
            u, st = p.controls()
            out.append((u, st, p.num_iter()))
            p.update_horizon(d["refs"][b, N + 1 + k])
        if b < dumped:
            assert_same(dumps[b][0], p.stages(), "stage records, instance %d" % b)
            assert_same_state(dumps[b][1], p.globals_(), "solver state, instance %d" % b)
        return out
    with ThreadPoolExecutor(8) as pool:       # the port's calls release the interpreter lock
        exp = list(pool.map(one, range(B)))
    bad = [(b, k) for b in range(B) for k in range(steps)
           if not (same_bits(got[k][0][b], exp[b][k][0]).all() and got[k][1][b] == exp[b][k][1]
                   and got[k][2][b] == exp[b][k][2])]
    assert not bad, bad[:10]
    return sum(int((g[1] == 0).sum()) for g in got)


@pytest.mark.parametrize("B,N,steps,seed", [(1024, 100, 5, 4242), (96, 500, 3, 4343),
                                            (777, 30, 4, 4444), (6401, 20, 3, 4545)])
def test_warm_control_steps_of_a_batch(port_built, B, N, steps, seed):
    """Warm steps are where half of the instances go through the handed-over
    step-length searches (the regrouped first attempt and the row-ordered
    continuation; at N = 500 their long-horizon forms).  From 512 instances on a
    Newton iteration runs as two parts on two streams, from 6 145 on the Newton
    solve takes its instances from the sorted list: both with instance counts
    that are no multiple of a wavefront's four."""
    converged = batch_against_port(B, N, steps, seed, dumped=24)
    assert converged > B * steps // 4        # the comparison is not vacuous


def test_hostile_inputs_through_the_solver(port_built):
    """Zero, huge, infinite and NaN entries in the measurements, the wind and the
    reference points of four instances in ten: every kernel ends, and controls,
    status, iteration counts and the complete end state are the port's (a NaN
    for a NaN)."""
    def mutate(d):
        rng = np.random.default_rng(99)
        special = np.array([0.0, -0.0, 1e-300, 1e150, -1e200, np.inf, -np.inf, np.nan, 1e6, -1e4])
        B = d["wind"].shape[0]
        pick = rng.random(B) < 0.4
        for key in ("meas", "wind", "refs"):
            arr = d[key]
            m = (rng.random(arr.shape) < 0.02) & pick.reshape((B,) + (1,) * (arr.ndim - 1))
            arr[m] = special[rng.integers(0, len(special), int(m.sum()))]
    converged = batch_against_port(160, 100, 3, 4444, dumped=160, mutate=mutate)
    assert 160 * 3 * 3 // 4 > converged > 160 * 3 // 4      # some instances fail on their inputs, the others solve


def test_one_part_iterations_in_a_fresh_process(port_built):
    """NMPC_ONE_PART keeps every Newton iteration on the batch's one stream (the
    sequence batches below 512 instances always take; above 6 144 instances with
    the listed Newton solve): same results, checked by the soak tool in a child
    process because the switch is read once per process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NMPC_ONE_PART="1", SOAK_DUMPS="8")
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "soak_warm.py"),
                          "777", "30", "3", "5151", "6401", "20", "2", "5152"],
                         env=env, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if "mismatches" in l]
    assert len(lines) == 2, out.stdout[-2000:]
    for l in lines:
        assert " 0 mismatches" in l and "0 entries differ" in l, l


def test_one_wavefront_sweep_in_a_fresh_process(port_built):
    """NMPC_NO_TEAM keeps the Newton solve's bottom-up sweep in one wavefront
    where the library otherwise runs it as a team of two (batches of up to 2 048
    instances and the long-sweep part of two-part iterations): same results,
    checked by the soak tool in a child process because the switch is read once
    per process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NMPC_NO_TEAM="1", SOAK_DUMPS="8")
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "soak_warm.py"),
                          "777", "30", "3", "5151", "96", "100", "3", "5153"],
                         env=env, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if "mismatches" in l]
    assert len(lines) == 2, out.stdout[-2000:]
    for l in lines:
        assert " 0 mismatches" in l and "0 entries differ" in l, l


@pytest.mark.gpu
def test_long_horizon_forms_in_fresh_processes(port_built):
    """From 200 stages on the Newton step leaves the Hessian block rows of the next
    set-up to a kernel of their own (several workgroups per instance), and the long
    part of a two-part iteration (or a batch of at most 256 instances) runs its
    stage kernels with workgroups of 1 024 threads instead of 256.
    NMPC_HESSIAN_APART and NMPC_WIDE_STAGE = 1 / 0 make every launch take those
    ways / none, and NMPC_FEW_SEARCHES=0 keeps a launch's step-length searches with
    the kernels for many however few they are: same results either way, at a short horizon with two-part
    iterations and at a long one, checked by the soak tool in child processes
    because the switches are read once per process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for forced, runs in (("1", ["777", "30", "3", "5151", "24", "100", "2", "5154"]),
                         ("0", ["20", "230", "2", "5155"]), ("1", ["20", "230", "2", "5155"])):
        env = dict(os.environ, NMPC_HESSIAN_APART=forced, NMPC_WIDE_STAGE=forced, NMPC_FEW_SEARCHES=forced,
                   SOAK_DUMPS="8")
        out = subprocess.run([sys.executable, os.path.join(root, "tests", "soak_warm.py")] + runs,
                             env=env, capture_output=True, text=True, timeout=170)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if "mismatches" in l]
        assert len(lines) == len(runs) // 4, out.stdout[-2000:]
        for l in lines:
            assert " 0 mismatches" in l and "0 entries differ" in l, l
