"""Test infrastructure only: ctypes driver for oracle/_ref/libref_n<N>.so.

The reference library (ccs-c66x/cnmpc.c, compiled by oracle/Makefile) keeps all
state in file statics that are never reset, so every scenario loads a private
copy of the shared object (copied to a temporary file first: dlopen of the same
path would hand back the already-initialised image).
"""
import ctypes as C
import os
import shutil
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
STAGE_REC = 332

STAGE_FIELDS = {  # name: (offset, length) inside a stage record
    "C": (0, 180), "c": (180, 12), "zLow": (192, 15), "zUpp": (207, 15),
    "q": (222, 15), "p": (237, 1), "z": (238, 15), "y": (253, 30),
    "qStep": (283, 15), "pStep": (298, 1), "zUnc": (299, 15), "dz": (314, 15),
}


def global_fields(n):
    out, off = {}, 0
    for name, ln in (("lambda", 12 * n), ("dLambda", 12 * n), ("grad", 12 * n),
                     ("hess", 288 * n), ("chol", 288 * n),
                     ("ieq", 15 * (n + 1)), ("prevIeq", 15 * (n + 1)),
                     ("xref", 13 * (n + 1)), ("uref", 3 * n), ("u", 3)):
        out[name] = (off, ln)
        off += ln
    return out, off


def ref_path(n=100):
    return os.path.join(HERE, "_ref", "libref_n%d.so" % n)


def have_ref(n=100):
    return os.path.exists(ref_path(n))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Ref:
    """One private instance of the reference controller."""

    def __init__(self, n=100):
        src = ref_path(n)
        if not os.path.exists(src):
            raise FileNotFoundError(src + " (run `make -C oracle ref N=%d`)" % n)
        fd, self._tmp = tempfile.mkstemp(suffix=".so", prefix="libref_")
        os.close(fd)
        shutil.copyfile(src, self._tmp)
        self.lib = lib = C.CDLL(self._tmp)
        os.unlink(self._tmp)
        self.N = n
        assert lib.ref_horizon() == n
        dp = C.POINTER(C.c_double)
        for name in ("nmpc_feedback_step", "nmpc_update_horizon",
                     "nmpc_set_state_weights", "nmpc_set_control_weights",
                     "nmpc_set_terminal_weights", "nmpc_set_lower_control_bound",
                     "nmpc_set_upper_control_bound"):
            getattr(lib, name).argtypes = [dp]
            getattr(lib, name).restype = None
        lib.nmpc_get_controls.argtypes = [dp]
        lib.nmpc_get_controls.restype = C.c_int
        lib.nmpc_set_reference_point.argtypes = [dp, C.c_uint32]
        lib.nmpc_set_reference_point.restype = None
        lib.nmpc_set_wind_velocity.argtypes = [C.c_double] * 3
        lib.nmpc_set_wind_velocity.restype = None
        lib.nmpc_init.restype = None
        lib.nmpc_preparation_step.restype = None
        lib.ref_alpha.restype = C.c_double
        lib.ref_get_stage.argtypes = [C.c_uint, dp]
        lib.ref_get_global.argtypes = [dp]
        lib.ref_rk4.argtypes = [dp, dp, dp, C.c_double]
        for name, na in (("ref_set_wind", 1), ("ref_x8_dynamics", 3),
                         ("ref_state_model", 3), ("ref_state_to_delta", 3),
                         ("ref_solve_interval_ivp", 5)):
            getattr(lib, name).argtypes = [dp] * na
            getattr(lib, name).restype = None
        self.gfields, self.gsize = global_fields(n)
        assert lib.ref_global_size() == self.gsize

    # -- controller API (c/cnmpc.h) ------------------------------------
    def setup(self, state_w, control_w, terminal_w, upper, lower):
        f = lambda v: _dp(np.ascontiguousarray(v, dtype=np.float64))
        self.lib.nmpc_set_state_weights(f(state_w))
        self.lib.nmpc_set_control_weights(f(control_w))
        self.lib.nmpc_set_terminal_weights(f(terminal_w))
        self.lib.nmpc_set_lower_control_bound(f(lower))
        self.lib.nmpc_set_upper_control_bound(f(upper))

    def init(self):
        self.lib.nmpc_init()

    def set_reference(self, point, i):
        p = np.ascontiguousarray(point, dtype=np.float64)
        assert p.shape == (16,)
        self.lib.nmpc_set_reference_point(_dp(p), i)

    def set_wind(self, w):
        self.lib.nmpc_set_wind_velocity(float(w[0]), float(w[1]), float(w[2]))

    def feedback(self, meas):
        m = np.ascontiguousarray(meas, dtype=np.float64)
        assert m.shape == (13,)
        self.lib.nmpc_feedback_step(_dp(m))

    def controls(self):
        u = np.zeros(3)
        st = self.lib.nmpc_get_controls(_dp(u))
        return u, st

    def update_horizon(self, point):
        p = np.ascontiguousarray(point, dtype=np.float64)
        assert p.shape == (16,)
        self.lib.nmpc_update_horizon(_dp(p))

    def num_iter(self):
        return self.lib.ref_num_iter()

    # -- internals ------------------------------------------------------
    def stages(self):
        out = np.zeros((self.N + 1, STAGE_REC))
        for k in range(self.N + 1):
            self.lib.ref_get_stage(k, _dp(out[k]))
        return out

    def globals_(self):
        out = np.zeros(self.gsize)
        self.lib.ref_get_global(_dp(out))
        return out

    def gfield(self, g, name):
        o, ln = self.gfields[name]
        return g[o:o + ln]

    # -- model functions ------------------------------------------------
    def x8_dynamics(self, x, u, wind):
        x, u, w = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, u, wind))
        out = np.zeros(6)
        self.lib.ref_set_wind(_dp(w))
        self.lib.ref_x8_dynamics(_dp(out), _dp(x), _dp(u))
        return out

    def state_model(self, x, u, wind):
        x, u, w = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, u, wind))
        out = np.zeros(13)
        self.lib.ref_set_wind(_dp(w))
        self.lib.ref_state_model(_dp(out), _dp(x), _dp(u))
        return out

    def rk4(self, x, u, wind, delta=1.0 / 50.0):
        x, u, w = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, u, wind))
        out = np.zeros(13)
        self.lib.ref_set_wind(_dp(w))
        self.lib.ref_rk4(_dp(out), _dp(x), _dp(u), delta)
        return out

    def state_to_delta(self, s1, s2):
        s1, s2 = (np.ascontiguousarray(a, dtype=np.float64) for a in (s1, s2))
        out = np.zeros(12)
        self.lib.ref_state_to_delta(_dp(out), _dp(s1), _dp(s2))
        return out

    def linearise(self, x, u, xnext, wind, lower, upper):
        """C (12x15 row-major) and c for one stage; sets the library's bounds."""
        x, u, xn, w = (np.ascontiguousarray(a, dtype=np.float64)
                       for a in (x, u, xnext, wind))
        lo = np.ascontiguousarray(lower, dtype=np.float64)
        up = np.ascontiguousarray(upper, dtype=np.float64)
        self.lib.nmpc_set_lower_control_bound(_dp(lo))
        self.lib.nmpc_set_upper_control_bound(_dp(up))
        self.lib.ref_set_wind(_dp(w))
        jac, res = np.zeros(180), np.zeros(12)
        self.lib.ref_solve_interval_ivp(_dp(x), _dp(u), _dp(xn), _dp(jac), _dp(res))
        return jac, res
