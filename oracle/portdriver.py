"""Test infrastructure only: ctypes driver for oracle/libnmpc_port.so (the CPU
restatement of the reference algorithm).  Same surface as refdriver.Ref so the
two can be run side by side; unlike the reference it is instance-based and the
horizon is a run-time parameter."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
STAGE_REC = 332
LIB_PATH = os.path.join(HERE, "libnmpc_port.so")
# builds of the port: the two precisions, and (double precision) the Heun and
# Euler integrators of the reference's compile-time matrix
LIB_PATHS = {"double": LIB_PATH, "float": os.path.join(HERE, "libnmpc_port_f32.so"),
             "heun": os.path.join(HERE, "libnmpc_port_heun.so"),
             "euler": os.path.join(HERE, "libnmpc_port_euler.so")}
DTYPES = {"double": np.float64, "float": np.float32, "heun": np.float64, "euler": np.float64}
CTYPES = {"double": C.c_double, "float": C.c_float, "heun": C.c_double, "euler": C.c_double}

_libs = {}


def _dp(a):
    ct = C.c_double if a.dtype == np.float64 else C.c_float
    return a.ctypes.data_as(C.POINTER(ct))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _f(v, n=None, precision="double"):
    a = np.ascontiguousarray(v, dtype=DTYPES[precision])
    if n is not None:
        assert a.size == n, (a.shape, n)
    return a


def lib(precision="double"):
    """The port built for `precision`: "double", or "float" (the reference's
    shipped single-precision configuration)."""
    if precision not in _libs:
        path = LIB_PATHS[precision]
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (run `make -C oracle port`)")
        L = C.CDLL(path)
        real = CTYPES[precision]
        assert L.port_real_bytes() == C.sizeof(real)
        dp, vp = C.POINTER(real), C.c_void_p
        L.port_create.restype = vp
        L.port_create.argtypes = [C.c_int]
        L.port_destroy.argtypes = [vp]
        for n in ("port_set_state_weights", "port_set_control_weights",
                  "port_set_terminal_weights", "port_set_lower_control_bound",
                  "port_set_upper_control_bound", "port_feedback_step",
                  "port_update_horizon", "port_get_global"):
            getattr(L, n).argtypes = [vp, dp]
            getattr(L, n).restype = None
        L.port_get_controls.argtypes = [vp, dp]
        L.port_set_wind_velocity.argtypes = [vp] + [real] * 3
        L.port_set_wind_velocity.restype = None
        L.port_init.argtypes = [vp]
        L.port_init.restype = None
        L.port_set_reference_point.argtypes = [vp, dp, C.c_uint]
        L.port_set_reference_point.restype = None
        for n in ("port_horizon", "port_num_iter", "port_last_result",
                  "port_last_status", "port_line_search_iter",
                  "port_hessian_regularized"):
            getattr(L, n).argtypes = [vp]
        L.port_alpha.argtypes = [vp]
        L.port_alpha.restype = real
        L.port_get_stage.argtypes = [vp, C.c_uint, dp]
        L.port_global_size.argtypes = [vp]
        L.port_global_size.restype = C.c_uint
        L.port_x8_dynamics.argtypes = [dp] * 4
        L.port_state_model.argtypes = [dp] * 4
        L.port_rk4.argtypes = [dp] * 4 + [real]
        L.port_state_to_delta.argtypes = [dp] * 3
        L.port_linearise.argtypes = [dp] * 8
        L.port_batch_cold_solve.restype = C.c_double
        L.port_batch_cold_solve.argtypes = (
            [C.c_int] * 3 + [dp] * 8 + [dp, C.POINTER(C.c_int), C.POINTER(C.c_int)])
        L.port_preparation_step_rti.argtypes = [vp]
        L.port_preparation_step_rti.restype = None
        L.port_batch_linearise.restype = C.c_double
        L.port_batch_linearise.argtypes = [C.c_long, C.c_int] + [dp] * 8
        _libs[precision] = L
    return _libs[precision]


def have_port(precision="double"):
    return os.path.exists(LIB_PATHS[precision])


class Port:
    def __init__(self, n=100, precision="double"):
        self.lib = lib(precision)
        self.precision = precision
        self.dtype = DTYPES[precision]
        self.N = n
        self.h = self.lib.port_create(n)
        from refdriver import global_fields
        self.gfields, self.gsize = global_fields(n)
        assert self.lib.port_global_size(self.h) == self.gsize

    def __del__(self):
        try:
            self.lib.port_destroy(self.h)
        except Exception:
            pass

    def setup(self, state_w, control_w, terminal_w, upper, lower):
        self.lib.port_set_state_weights(self.h, _dp(_f(state_w, 12, precision=self.precision)))
        self.lib.port_set_control_weights(self.h, _dp(_f(control_w, 3, precision=self.precision)))
        self.lib.port_set_terminal_weights(self.h, _dp(_f(terminal_w, 12, precision=self.precision)))
        self.lib.port_set_lower_control_bound(self.h, _dp(_f(lower, 3, precision=self.precision)))
        self.lib.port_set_upper_control_bound(self.h, _dp(_f(upper, 3, precision=self.precision)))

    def init(self):
        self.lib.port_init(self.h)

    def set_reference(self, point, i):
        self.lib.port_set_reference_point(self.h, _dp(_f(point, 16, precision=self.precision)), i)

    def set_wind(self, w):
        self.lib.port_set_wind_velocity(self.h, float(w[0]), float(w[1]), float(w[2]))

    def feedback(self, meas):
        self.lib.port_feedback_step(self.h, _dp(_f(meas, 13, precision=self.precision)))

    def controls(self):
        u = np.zeros(3, dtype=self.dtype)
        st = self.lib.port_get_controls(self.h, _dp(u))
        return u, st

    def prepare_rti(self):
        self.lib.port_preparation_step_rti(self.h)

    def update_horizon(self, point):
        self.lib.port_update_horizon(self.h, _dp(_f(point, 16, precision=self.precision)))

    def num_iter(self):
        return self.lib.port_num_iter(self.h)

    def stages(self):
        out = np.zeros((self.N + 1, STAGE_REC), dtype=self.dtype)
        for k in range(self.N + 1):
            self.lib.port_get_stage(self.h, k, _dp(out[k]))
        return out

    def globals_(self):
        out = np.zeros(self.gsize, dtype=self.dtype)
        self.lib.port_get_global(self.h, _dp(out))
        return out

    def gfield(self, g, name):
        o, ln = self.gfields[name]
        return g[o:o + ln]


# -- model functions ------------------------------------------------------
def x8_dynamics(x, u, wind, precision="double"):
    out = np.zeros(6, dtype=DTYPES[precision])
    lib(precision).port_x8_dynamics(_dp(out), _dp(_f(x, 13, precision=precision)), _dp(_f(u, 3, precision=precision)), _dp(_f(wind, 3, precision=precision)))
    return out


def state_model(x, u, wind, precision="double"):
    out = np.zeros(13, dtype=DTYPES[precision])
    lib(precision).port_state_model(_dp(out), _dp(_f(x, 13, precision=precision)), _dp(_f(u, 3, precision=precision)), _dp(_f(wind, 3, precision=precision)))
    return out


def rk4(x, u, wind, delta=1.0 / 50.0, precision="double"):
    out = np.zeros(13, dtype=DTYPES[precision])
    lib(precision).port_rk4(_dp(out), _dp(_f(x, 13, precision=precision)), _dp(_f(u, 3, precision=precision)), _dp(_f(wind, 3, precision=precision)), delta)
    return out


def state_to_delta(s1, s2, precision="double"):
    out = np.zeros(12, dtype=DTYPES[precision])
    lib(precision).port_state_to_delta(_dp(out), _dp(_f(s1, 13, precision=precision)), _dp(_f(s2, 13, precision=precision)))
    return out


def linearise(x, u, xnext, wind, lower, upper, precision="double"):
    jac, res = np.zeros(180, dtype=DTYPES[precision]), np.zeros(12, dtype=DTYPES[precision])
    lib(precision).port_linearise(_dp(_f(x, 13, precision=precision)), _dp(_f(u, 3, precision=precision)), _dp(_f(xnext, 13, precision=precision)),
                         _dp(_f(wind, 3, precision=precision)), _dp(_f(lower, 3, precision=precision)), _dp(_f(upper, 3, precision=precision)),
                         _dp(jac), _dp(res))
    return jac, res


def batch_cold_solve(horizon, wq, wr, wt, lower, upper, refs, meas, wind, threads=1,
                     precision="double"):
    """refs [B][N+1][16], meas [B][13], wind [B][3] -> (u, status, iters, seconds)"""
    refs, meas, wind = _f(refs, precision=precision), _f(meas, precision=precision), _f(wind, precision=precision)
    B = meas.shape[0]
    assert refs.shape == (B, horizon + 1, 16) and wind.shape == (B, 3)
    u = np.zeros((B, 3), dtype=DTYPES[precision])
    status = np.zeros(B, dtype=np.int32)
    iters = np.zeros(B, dtype=np.int32)
    secs = lib(precision).port_batch_cold_solve(
        horizon, B, threads, _dp(_f(wq, 12, precision=precision)), _dp(_f(wr, 3, precision=precision)), _dp(_f(wt, 12, precision=precision)),
        _dp(_f(lower, 3, precision=precision)), _dp(_f(upper, 3, precision=precision)), _dp(refs), _dp(meas), _dp(wind),
        _dp(u), _ip(status), _ip(iters))
    return u, status, iters, secs


def batch_linearise(x, u, xnext, wind, lower, upper, threads=1, precision="double"):
    """[n][13], [n][3], [n][13], [n][3] -> (jac [n][180], res [n][12], seconds)"""
    x, u, xnext, wind = _f(x, precision=precision), _f(u, precision=precision), _f(xnext, precision=precision), _f(wind, precision=precision)
    n = x.shape[0]
    jac, res = np.zeros((n, 180), dtype=DTYPES[precision]), np.zeros((n, 12), dtype=DTYPES[precision])
    secs = lib(precision).port_batch_linearise(n, threads, _dp(x), _dp(u), _dp(xnext), _dp(wind),
                                      _dp(_f(lower, 3, precision=precision)), _dp(_f(upper, 3, precision=precision)), _dp(jac), _dp(res))
    return jac, res, secs
