"""GPU parity test (pytest -m gpu) of the real-time-iteration preparation step:
nmpc_batch_set_rti + nmpc_batch_preparation_step against the CPU oracle port
extended the same way (oracle/nmpc_port.c: port_preparation_step_rti).  The
reference's own preparation step is empty, so this mode has no reference output
to compare with: parity is against the port only."""
import numpy as np
import pytest

from helpers import assert_same, assert_same_state

pytestmark = pytest.mark.gpu


def test_rti_preparation_matches_the_port(port_built):
    import portdriver as P
    from nmpc_amd import synthetic as S
    from nmpc_amd.batch import Batch
    N, B, steps = 30, 24, 8
    d = S.make_batch(B, N, extra_steps=steps, seed=616, turn_start=(0.05, 0.4))
    cfg = S.setup_args()

    def gpu(rti):
        bt = Batch(B, N)
        bt.setup(**cfg)
        bt.set_rti(rti)
        bt.initialise_horizon()
        bt.set_wind_velocity(d["wind"])
        bt.set_reference(d["refs"][:, :N + 1])
        out = []
        for k in range(steps):
            bt.prepare()
            if k == 3:
                bt.prepare()        # a second call without a solve in between changes nothing
            bt.solve(d["meas"][:, k])
            out.append(bt.get_controls())
            bt.update_horizon(d["refs"][:, N + 1 + k])
        st = [bt.stages(b) for b in range(B)]
        gl = [bt.globals_(b) for b in range(B)]
        bt.close()
        return out, st, gl

    got, gst, ggl = gpu(True)
    plain, _, _ = gpu(False)
    differs = False
    # both kinds of instance must occur: after a converged solve everything is
    # linearised again, after an unconverged one the instance is left alone
    status = np.stack([got[k][1] for k in range(steps)])
    assert (status[:-1] == 0).any() and (status[:-1] != 0).any(), status
    for b in range(B):
        c = P.Port(N)
        c.setup(cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"],
                cfg["upper_control_bound"], cfg["lower_control_bound"])
        c.init()
        c.set_wind(d["wind"][b])
        for i in range(N + 1):
            c.set_reference(d["refs"][b, i], i)
        for k in range(steps):
            c.prepare_rti()
            c.feedback(d["meas"][b, k])
            u, st = c.controls()
            assert_same(got[k][0][b], u, "controls, instance %d step %d" % (b, k))
            assert got[k][1][b] == st and got[k][2][b] == c.num_iter(), (b, k)
            differs = differs or not np.array_equal(got[k][0][b], plain[k][0][b])
            c.update_horizon(d["refs"][b, N + 1 + k])
        assert_same(gst[b], c.stages(), "stage records, instance %d" % b)
        assert_same_state(ggl[b], c.globals_(), "solver state, instance %d" % b)
    assert differs, "the mode must change the linearisation (and so the controls) after the first step"
    # the first step has nothing to linearise about: same as the plain mode
    assert_same(got[0][0], plain[0][0], "first step")
