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


def ref_path(n=100, precision="double"):
    tag = "n%d" % n if precision == "double" else "f32_n%d" % n
    return os.path.join(HERE, "_ref", "libref_%s.so" % tag)


def have_ref(n=100, precision="double"):
    return os.path.exists(ref_path(n, precision))


class Ref:
    """One private instance of the reference controller."""

    def __init__(self, n=100, precision="double"):
        """precision "double": the fp64 configuration; "float": the reference's
        shipped single-precision configuration.  All arrays in and out are
        numpy arrays of that type."""
        src = ref_path(n, precision)
        if not os.path.exists(src):
            raise FileNotFoundError(src + " (run `make -C oracle ref N=%d PREC=%s`)" % (n, precision))
        self.dtype = np.float64 if precision == "double" else np.float32
        self.ctype = C.c_double if precision == "double" else C.c_float
        fd, self._tmp = tempfile.mkstemp(suffix=".so", prefix="libref_")
        os.close(fd)
        shutil.copyfile(src, self._tmp)
        self.lib = lib = C.CDLL(self._tmp)
        os.unlink(self._tmp)
        self.N = n
        assert lib.ref_horizon() == n
        assert lib.ref_real_bytes() == np.dtype(self.dtype).itemsize
        dp = C.POINTER(self.ctype)
        real = self.ctype
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
        lib.nmpc_set_wind_velocity.argtypes = [real] * 3
        lib.nmpc_set_wind_velocity.restype = None
        lib.nmpc_init.restype = None
        lib.nmpc_preparation_step.restype = None
        lib.ref_alpha.restype = real
        lib.ref_get_stage.argtypes = [C.c_uint, dp]
        lib.ref_get_global.argtypes = [dp]
        lib.ref_rk4.argtypes = [dp, dp, dp, real]
        for name, na in (("ref_set_wind", 1), ("ref_x8_dynamics", 3),
                         ("ref_state_model", 3), ("ref_state_to_delta", 3),
                         ("ref_solve_interval_ivp", 5)):
            getattr(lib, name).argtypes = [dp] * na
            getattr(lib, name).restype = None
        self.gfields, self.gsize = global_fields(n)
        assert lib.ref_global_size() == self.gsize

    def _a(self, v, n=None):
        a = np.ascontiguousarray(v, dtype=self.dtype)
        if n is not None:
            assert a.shape == (n,), (a.shape, n)
        return a

    def _p(self, a):
        assert a.dtype == self.dtype
        return a.ctypes.data_as(C.POINTER(self.ctype))

    # -- controller API (c/cnmpc.h) ------------------------------------
    def setup(self, state_w, control_w, terminal_w, upper, lower):
        f = lambda v: self._p(self._a(v))
        self.lib.nmpc_set_state_weights(f(state_w))
        self.lib.nmpc_set_control_weights(f(control_w))
        self.lib.nmpc_set_terminal_weights(f(terminal_w))
        self.lib.nmpc_set_lower_control_bound(f(lower))
        self.lib.nmpc_set_upper_control_bound(f(upper))

    def init(self):
        self.lib.nmpc_init()

    def set_reference(self, point, i):
        self.lib.nmpc_set_reference_point(self._p(self._a(point, 16)), i)

    def set_wind(self, w):
        self.lib.nmpc_set_wind_velocity(float(w[0]), float(w[1]), float(w[2]))

    def feedback(self, meas):
        self.lib.nmpc_feedback_step(self._p(self._a(meas, 13)))

    def controls(self):
        u = np.zeros(3, dtype=self.dtype)
        st = self.lib.nmpc_get_controls(self._p(u))
        return u, st

    def update_horizon(self, point):
        self.lib.nmpc_update_horizon(self._p(self._a(point, 16)))

    def num_iter(self):
        return self.lib.ref_num_iter()

    # -- internals ------------------------------------------------------
    def stages(self):
        out = np.zeros((self.N + 1, STAGE_REC), dtype=self.dtype)
        for k in range(self.N + 1):
            self.lib.ref_get_stage(k, self._p(out[k]))
        return out

    def globals_(self):
        out = np.zeros(self.gsize, dtype=self.dtype)
        self.lib.ref_get_global(self._p(out))
        return out

    def gfield(self, g, name):
        o, ln = self.gfields[name]
        return g[o:o + ln]

    # -- model functions ------------------------------------------------
    def x8_dynamics(self, x, u, wind):
        x, u, w = (self._a(a) for a in (x, u, wind))
        out = np.zeros(6, dtype=self.dtype)
        self.lib.ref_set_wind(self._p(w))
        self.lib.ref_x8_dynamics(self._p(out), self._p(x), self._p(u))
        return out

    def state_model(self, x, u, wind):
        x, u, w = (self._a(a) for a in (x, u, wind))
        out = np.zeros(13, dtype=self.dtype)
        self.lib.ref_set_wind(self._p(w))
        self.lib.ref_state_model(self._p(out), self._p(x), self._p(u))
        return out

    def rk4(self, x, u, wind, delta=1.0 / 50.0):
        x, u, w = (self._a(a) for a in (x, u, wind))
        out = np.zeros(13, dtype=self.dtype)
        self.lib.ref_set_wind(self._p(w))
        self.lib.ref_rk4(self._p(out), self._p(x), self._p(u), delta)
        return out

    def state_to_delta(self, s1, s2):
        s1, s2 = (self._a(a) for a in (s1, s2))
        out = np.zeros(12, dtype=self.dtype)
        self.lib.ref_state_to_delta(self._p(out), self._p(s1), self._p(s2))
        return out

    def linearise(self, x, u, xnext, wind, lower, upper):
        """C (12x15 row-major) and c for one stage; sets the library's bounds."""
        x, u, xn, w = (self._a(a)
                       for a in (x, u, xnext, wind))
        lo = self._a(lower)
        up = self._a(upper)
        self.lib.nmpc_set_lower_control_bound(self._p(lo))
        self.lib.nmpc_set_upper_control_bound(self._p(up))
        self.lib.ref_set_wind(self._p(w))
        jac, res = np.zeros(180, dtype=self.dtype), np.zeros(12, dtype=self.dtype)
        self.lib.ref_solve_interval_ivp(self._p(x), self._p(u), self._p(xn), self._p(jac), self._p(res))
        return jac, res
