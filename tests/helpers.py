"""Shared test plumbing: golden fixtures and sequence drivers for the CPU
oracle port and for the HIP library."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SEQUENCES = ["seq_straight", "seq_turn0", "seq_turn1", "seq_turn2", "seq_turn3",
             "seq_turn4", "seq_turn5", "seq_bound", "seq_identity", "seq_regularised",
             "seq_linesearch", "seq_exact", "seq_indefinite", "seq_n500_0", "seq_n500_1"]


# the same from the reference in its shipped single-precision configuration
SEQUENCES_F32 = ["seq_straight_f32", "seq_turn0_f32", "seq_turn1_f32", "seq_turn2_f32",
                 "seq_turn3_f32", "seq_bound_f32", "seq_identity_f32"]


def precision_of(g):
    """"float" for a fixture of the single-precision reference, else "double"."""
    return "float" if g["u"].dtype == np.float32 else "double"


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def config_of(g):
    return dict(state_weights=g["state_weights"], control_weights=g["control_weights"],
                terminal_weights=g["terminal_weights"], upper_control_bound=g["upper"],
                lower_control_bound=g["lower"])


def drive_port(g, dump=False):
    """Replay a golden sequence through the CPU oracle port."""
    import portdriver as P
    N = int(g["horizon"])
    steps = len(g["u"])
    c = config_of(g)
    p = P.Port(N, precision_of(g))
    dt = g["u"].dtype
    p.setup(c["state_weights"], c["control_weights"], c["terminal_weights"],
            c["upper_control_bound"], c["lower_control_bound"])
    p.init()
    p.set_wind(g["winds"][0])
    for i in range(N + 1):
        p.set_reference(g["refs"][i], i)
    out = dict(u=np.zeros((steps, 3), dt), status=np.zeros(steps, np.int32),
               iters=np.zeros(steps, np.int32), z0=np.zeros((steps, 15), dt))
    if dump:
        out["loaded_Cc"] = p.stages()[:, :192].copy()
    for k in range(steps):
        p.set_wind(g["winds"][k])
        p.feedback(g["meas"][k])
        out["u"][k], out["status"][k] = p.controls()
        out["iters"][k] = p.num_iter()
        out["z0"][k] = p.stages()[0, 238:253]
        p.update_horizon(g["refs"][N + 1 + k])
    if dump:
        out["final_stages"] = p.stages()
        out["final_globals"] = p.globals_()
    return out


def drive_gpu_batch(gs, dump=False):
    """Replay golden sequences of equal horizon, step count and configuration
    as one batch through libcnmpc.so (batch interface)."""
    from nmpc_amd.batch import Batch
    g0 = gs[0]
    N = int(g0["horizon"])
    steps = len(g0["u"])
    B = len(gs)
    bt = Batch(B, N)
    bt.setup(**config_of(g0))
    bt.initialise_horizon()
    bt.set_wind_velocity(np.stack([g["winds"][0] for g in gs]))
    bt.set_reference(np.stack([g["refs"][:N + 1] for g in gs]))
    dt = bt.dtype
    out = dict(u=np.zeros((B, steps, 3), dt), status=np.zeros((B, steps), np.int32),
               iters=np.zeros((B, steps), np.int32), z0=np.zeros((B, steps, 15), dt))
    if dump:
        out["loaded_Cc"] = [bt.stages(b)[:, :192].copy() for b in range(B)]
    for k in range(steps):
        bt.prepare()
        bt.set_wind_velocity(np.stack([g["winds"][k] for g in gs]))
        bt.solve(np.stack([g["meas"][k] for g in gs]))
        u, st, it = bt.get_controls()
        out["u"][:, k], out["status"][:, k], out["iters"][:, k] = u, st, it
        for b in range(B):
            rec = np.zeros(332, dt)
            bt._ck(bt.lib.nmpc_batch_get_stage(bt.h, b, 0, rec.ctypes.data_as(bt.lib.preal)))
            out["z0"][b, k] = rec[238:253]
        bt.update_horizon(np.stack([g["refs"][N + 1 + k] for g in gs]))
    if dump:
        out["final_stages"] = [bt.stages(b) for b in range(B)]
        out["final_globals"] = [bt.globals_(b) for b in range(B)]
    bt.close()
    return out


def drive_reinit(ctrl, g):
    """Replay the seq_reinit fixture (two initialisations of one controller in a
    row, see tests/golden/generate.py) through `ctrl`, an object with the
    surface of portdriver.Port, comparing as it goes."""
    N, steps = int(g["horizon"]), int(g["steps"])
    for ph in range(2):
        ctrl.setup(g["state_weights_%d" % ph], g["control_weights_%d" % ph], g["terminal_weights_%d" % ph],
                   g["upper_control_bound_%d" % ph], g["lower_control_bound_%d" % ph])
        ctrl.init()
        ctrl.set_wind(g["winds"][ph])
        for i in range(N + 1):
            ctrl.set_reference(g["refs"][ph, i], i)
        assert_same(ctrl.stages(), g["loaded_%d" % ph], "stage records after loading, phase %d" % ph)
        for k in range(steps):
            ctrl.feedback(g["meas"][ph, k])
            u, st = ctrl.controls()
            what = "phase %d step %d" % (ph, k)
            assert st == g["status"][ph, k], what
            assert ctrl.num_iter() == g["iters"][ph, k], what
            assert_same(u, g["u"][ph, k], "controls, " + what)
            ctrl.update_horizon(g["refs"][ph, N + 1 + k])
        assert_same(ctrl.stages(), g["final_stages_%d" % ph], "final stage records, phase %d" % ph)
        assert_same_state(ctrl.globals_(), g["final_globals_%d" % ph], "final solver state, phase %d" % ph)


class BatchAsController:
    """One instance of a nmpc_amd.batch.Batch of identical instances behind the
    surface of portdriver.Port."""

    def __init__(self, horizon, copies=1, watch=0):
        from nmpc_amd.batch import Batch
        self.bt, self.B, self.b = Batch(copies, horizon), copies, watch

    def _all(self, v):
        return np.repeat(np.asarray(v, dtype=self.bt.dtype)[None], self.B, axis=0)

    def setup(self, state_w, control_w, terminal_w, upper, lower):
        self.bt.setup(state_w, control_w, terminal_w, upper, lower)

    def init(self):
        self.bt.initialise_horizon()

    def set_wind(self, w):
        self.bt.set_wind_velocity(self._all(w))

    def set_reference(self, point, i):
        self.bt.set_reference(self._all(point)[:, None, :], first=i)

    def feedback(self, meas):
        self.bt.solve(self._all(meas))
        self.last = self.bt.get_controls()

    def controls(self):
        return self.last[0][self.b], int(self.last[1][self.b])

    def num_iter(self):
        return int(self.last[2][self.b])

    def update_horizon(self, point):
        self.bt.update_horizon(self._all(point))

    def stages(self):
        return self.bt.stages(self.b)

    def globals_(self):
        return self.bt.globals_(self.b)

    def close(self):
        self.bt.close()


def same_bits(a, b):
    """Element-wise: the same bit pattern (so -0.0 differs from +0.0), or both NaN
    (x86 and gfx950 produce default NaNs of opposite sign)."""
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind != "f" or b.dtype != a.dtype:
        return a == b
    bits = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return (a.view(bits) == b.view(bits)) | (np.isnan(a) & np.isnan(b))


def read_by_solver(g):
    """A solver-state dump (refdriver.global_fields) with the entries blanked that
    the reference computes and never reads again: right of the diagonal in the
    diagonal blocks of the Newton Hessian (the reverse Cholesky takes the entries
    (i, j <= i) and the sub-diagonal block only, qpDUNES/dual_qp.c:835, 871, 892).
    The HIP library neither computes nor stores them; its dump shows zeros there."""
    g = np.array(g, copy=True)
    n = (g.size - 46) // 658
    assert g.size == 658 * n + 46, g.size
    hess = g[36 * n:36 * n + 288 * n].reshape(n, 12, 24)
    l, j = np.meshgrid(np.arange(12), np.arange(12), indexing="ij")
    hess[:, l[j > l], 12 + j[j > l]] = 0
    return g


def assert_same_state(got, exp, what):
    """assert_same for two solver-state dumps, on what the algorithm reads of them."""
    assert_same(read_by_solver(got), read_by_solver(exp), what)


def assert_same(a, b, what):
    """Bit-for-bit equality with a readable first difference."""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    ok = same_bits(a, b)
    if not np.all(ok):
        d = np.argwhere(~ok)
        i = tuple(d[0])
        raise AssertionError("%s: %d of %d entries differ, first at %s: %r vs %r" % (
            what, len(d), a.size, i, a[i], b[i]))
