"""Python wrapper of the nmpc_batch_* interface: many independent NMPC
controllers in lockstep on one MI355X.  Method names follow the reference's
Python module (python/nmpc/__init__.py:51-92): setup, initialise_horizon,
set_reference, prepare, solve, get_controls, set_wind_velocity, update_horizon.

Inputs may be numpy arrays (host) or torch CUDA tensors (device, float64,
contiguous); device inputs are used in place without a copy.
"""
import ctypes as C

import numpy as np

from . import _lib


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class _Arg:
    """Pointer + keep-alive for a [..] input of the library's real type."""

    def __init__(self, x, shape, dtype=np.float64):
        if _is_torch(x):
            import torch
            want = torch.float64 if np.dtype(dtype) == np.float64 else torch.float32
            if x.dtype != want or not x.is_contiguous():
                raise TypeError("device inputs must be contiguous %s tensors" % np.dtype(dtype).name)
            if tuple(x.shape) != tuple(shape):
                raise ValueError("expected shape %s, got %s" % (shape, tuple(x.shape)))
            self.keep = x
            self.ptr = C.c_void_p(x.data_ptr())
        else:
            a = np.ascontiguousarray(x, dtype=dtype)
            if a.shape != tuple(shape):
                raise ValueError("expected shape %s, got %s" % (shape, a.shape))
            self.keep = a
            self.ptr = C.c_void_p(a.ctypes.data)


def _vec(v, n, dtype=np.float64):
    a = np.ascontiguousarray(v, dtype=dtype)
    if a.shape != (n,):
        raise ValueError("expected %d values" % n)
    return a


class Batch:
    def __init__(self, instances, horizon=100, device=0, stream=None):
        self.lib = _lib.load()
        self.dtype = np.dtype(self.lib.np_real)     # float64, or float32 for libcnmpc_f32.so
        self.B, self.N = int(instances), int(horizon)
        h = C.c_void_p()
        _lib.check(self.lib, self.lib.nmpc_batch_create(
            C.byref(h), self.B, self.N, int(device),
            C.c_void_p(stream) if stream else None))
        self.h = h
        self._own_stream = not stream
        self._stream_ptr = stream
        self._torch_stream = None       # the caller's stream as a torch object, made on first use
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.nmpc_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        _lib.check(self.lib, rc)

    def _arg(self, x, shape):
        """Input argument; a device tensor is kept alive until the next wait.  A
        batch on its own stream is not ordered behind the torch stream that
        produced the tensor, so that stream is drained first (pass torch's stream
        to Batch() to avoid the wait)."""
        a = _Arg(x, shape, self.dtype)
        if _is_torch(x):
            import torch
            if self._own_stream:
                torch.cuda.current_stream(x.device).synchronize()
                # the library's private stream is not torch's to know: the tensor is
                # kept here until the next wait (a list of 256 forces one)
                if len(self._keep) >= 256:
                    self.synchronize()
                self._keep.append(a)
            else:
                # a stream the caller gave: torch's allocator is told that the tensor is
                # in use on it and frees the memory only behind the work queued there,
                # with no wait and nothing kept here
                if self._torch_stream is None:
                    self._torch_stream = torch.cuda.ExternalStream(self._stream_ptr, device=x.device)
                x.record_stream(self._torch_stream)
        return a

    # -- configuration -------------------------------------------------
    def setup(self, state_weights, control_weights, terminal_weights,
              upper_control_bound, lower_control_bound):
        dp = lambda a: a.ctypes.data_as(self.lib.preal)
        _vec = lambda v, n: globals()["_vec"](v, n, self.dtype)
        self._ck(self.lib.nmpc_batch_set_state_weights(self.h, dp(_vec(state_weights, 12))))
        self._ck(self.lib.nmpc_batch_set_control_weights(self.h, dp(_vec(control_weights, 3))))
        self._ck(self.lib.nmpc_batch_set_terminal_weights(self.h, dp(_vec(terminal_weights, 12))))
        self._ck(self.lib.nmpc_batch_set_lower_control_bound(self.h, dp(_vec(lower_control_bound, 3))))
        self._ck(self.lib.nmpc_batch_set_upper_control_bound(self.h, dp(_vec(upper_control_bound, 3))))

    def setup_instances(self, first, state_weights=None, control_weights=None, terminal_weights=None,
                        upper_control_bound=None, lower_control_bound=None):
        """Weights and control bounds of instances first .. first+count-1 alone:
        arrays [count][12] / [count][3] (numpy or torch CUDA tensors), None leaves a
        setting as it is.  Weights take effect at the next initialise_horizon()."""
        given = [a for a in (state_weights, control_weights, terminal_weights,
                             upper_control_bound, lower_control_bound) if a is not None]
        if not given:
            return
        count = int(given[0].shape[0])
        arg = lambda a, n: self._arg(a, (count, n)) if a is not None else None
        ptr = lambda a: a.ptr if a is not None else None
        wq, wr, wt = arg(state_weights, 12), arg(control_weights, 3), arg(terminal_weights, 12)
        ub, lb = arg(upper_control_bound, 3), arg(lower_control_bound, 3)
        self._ck(self.lib.nmpc_batch_set_instance_weights(self.h, int(first), count, ptr(wq), ptr(wr), ptr(wt)))
        self._ck(self.lib.nmpc_batch_set_instance_control_bounds(self.h, int(first), count, ptr(lb), ptr(ub)))

    def initialise_horizon(self):
        self._ck(self.lib.nmpc_batch_init(self.h))

    def set_wind_velocity(self, wind):
        a = self._arg(wind, (self.B, 3))
        self._ck(self.lib.nmpc_batch_set_wind_velocity(self.h, a.ptr))

    def set_reference(self, points, first=0):
        """points [B][count][16] -> horizon positions first .. first+count-1."""
        shp = tuple(points.shape)
        if len(shp) != 3 or shp[0] != self.B or shp[2] != 16:
            raise ValueError("expected [B][count][16]")
        a = self._arg(points, shp)
        self._ck(self.lib.nmpc_batch_set_reference_points(self.h, a.ptr, int(first), shp[1]))

    def set_reference_from_waypoints(self, waypoints, t0, dt, first=0, count=None):
        """waypoints [B][W][17] rows (t, 16-vector), sorted by t: horizon positions
        first .. first+count-1 interpolated at t0 + position * dt on the device."""
        shp = tuple(waypoints.shape)
        if len(shp) != 3 or shp[0] != self.B or shp[2] != 17:
            raise ValueError("expected [B][W][17]")
        count = self.N + 1 - first if count is None else count
        a = self._arg(waypoints, shp)
        self._ck(self.lib.nmpc_batch_set_reference_from_waypoints(
            self.h, a.ptr, shp[1], float(t0), float(dt), int(first), int(count)))

    def set_reference_from_waypoints_noalloc(self, waypoints, t0, dt, first=0, count=None):
        """The same for callers that cannot allocate device memory: device
        waypoints only, at most NMPC_STATIC_SCRATCH_POINTS = 65536 points
        (instances x count) per call."""
        shp = tuple(waypoints.shape)
        if len(shp) != 3 or shp[0] != self.B or shp[2] != 17:
            raise ValueError("expected [B][W][17]")
        count = self.N + 1 - first if count is None else count
        a = self._arg(waypoints, shp)
        self._ck(self.lib.nmpc_batch_set_reference_from_waypoints_noalloc(
            self.h, a.ptr, shp[1], float(t0), float(dt), int(first), int(count)))

    # -- control loop --------------------------------------------------
    def prepare(self):
        self._ck(self.lib.nmpc_batch_preparation_step(self.h))

    def set_rti(self, enabled=True):
        """Real-time-iteration mode: prepare() re-linearises the stages about the
        last solution (see nmpc_batch_set_rti).  Off by default."""
        self._ck(self.lib.nmpc_batch_set_rti(self.h, 1 if enabled else 0))

    def relinearise(self):
        """Every stage linearised again about the stored references at the current
        wind estimate (see nmpc_batch_relinearise)."""
        self._ck(self.lib.nmpc_batch_relinearise(self.h))

    def solve(self, measurement):
        a = self._arg(measurement, (self.B, 13))
        self._ck(self.lib.nmpc_batch_feedback_step(self.h, a.ptr))

    def get_controls(self):
        """-> (controls [B][3], status [B] (0 ok / 1 infeasible), iterations [B])"""
        u = np.zeros((self.B, 3), dtype=self.dtype)
        st = np.zeros(self.B, dtype=np.int32)
        it = np.zeros(self.B, dtype=np.int32)
        self._ck(self.lib.nmpc_batch_get_controls(
            self.h, C.c_void_p(u.ctypes.data), C.c_void_p(st.ctypes.data),
            C.c_void_p(it.ctypes.data)))
        self._keep = []         # host outputs: the call waited for the stream
        return u, st, it

    def get_controls_into(self, controls, status, iterations):
        """Device-resident outputs (torch CUDA tensors: f64 [B][3], i32 [B], i32 [B]),
        written on the batch's stream; the call does not wait for them."""
        self._ck(self.lib.nmpc_batch_get_controls(
            self.h, C.c_void_p(controls.data_ptr()), C.c_void_p(status.data_ptr()),
            C.c_void_p(iterations.data_ptr())))

    def update_horizon(self, new_reference):
        a = self._arg(new_reference, (self.B, 16))
        self._ck(self.lib.nmpc_batch_update_horizon(self.h, a.ptr))

    # -- plant roll-forward (closed loop on the device) ---------------------
    def plant_set_state(self, states):
        a = self._arg(states, (self.B, 13))
        self._ck(self.lib.nmpc_batch_plant_set_state(self.h, a.ptr))

    def plant_state(self):
        x = np.zeros((self.B, 13), dtype=self.dtype)
        self._ck(self.lib.nmpc_batch_plant_get_state(self.h, x.ctypes.data_as(C.c_void_p)))
        return x

    def plant_integrate(self, dt, controls=None, wind=None):
        """One RK4 step of every simulated vehicle.  controls None: the last good
        controls of each controller; wind None: the controllers' wind estimate."""
        ac = self._arg(controls, (self.B, 3)) if controls is not None else None
        aw = self._arg(wind, (self.B, 3)) if wind is not None else None
        self._ck(self.lib.nmpc_batch_plant_integrate(
            self.h, float(dt), ac.ptr if ac else None, aw.ptr if aw else None))

    def solve_from_plant(self):
        self._ck(self.lib.nmpc_batch_feedback_step_from_plant(self.h))

    def reset(self):
        """Cold-start state again (see nmpc_batch_reset); weights and bounds stay."""
        self._ck(self.lib.nmpc_batch_reset(self.h))

    def synchronize(self):
        self._ck(self.lib.nmpc_batch_synchronize(self.h))
        self._keep = []

    # -- introspection ---------------------------------------------------
    def stages(self, instance):
        out = np.zeros((self.N + 1, 332), dtype=self.dtype)
        for k in range(self.N + 1):
            self._ck(self.lib.nmpc_batch_get_stage(
                self.h, instance, k, out[k].ctypes.data_as(self.lib.preal)))
        return out

    def globals_(self, instance):
        out = np.zeros(self.lib.nmpc_batch_global_size(self.h), dtype=self.dtype)
        self._ck(self.lib.nmpc_batch_get_global(self.h, instance, out.ctypes.data_as(self.lib.preal)))
        return out

    def solver_info(self, instance):
        info = (C.c_int32 * 6)()
        alpha = self.lib.real()
        self._ck(self.lib.nmpc_batch_get_solver_info(self.h, instance, info, C.byref(alpha)))
        keys = ("status", "num_iter", "last_change", "regularized", "ls_iter", "last_result")
        d = dict(zip(keys, list(info)))
        d["alpha"] = alpha.value
        return d

    # -- profiling -------------------------------------------------------
    def set_profiling(self, enabled=True, kinds=None):
        """Per-kernel timing on or off; kinds: names of the kernels to time (all if None)."""
        mask = 0
        for name in kinds or ():
            mask |= 1 << _lib.KERNEL_NAMES.index(name)
        self._ck(self.lib.nmpc_batch_set_profiling_kinds(self.h, mask))
        self._ck(self.lib.nmpc_batch_set_profiling(self.h, 1 if enabled else 0))

    def kernel_times(self):
        """{kernel name: (total ms, launches)} since profiling was enabled."""
        out = {}
        for i, name in enumerate(_lib.KERNEL_NAMES):
            ms, n = C.c_double(), C.c_uint64()
            self._ck(self.lib.nmpc_batch_get_kernel_time(self.h, i, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out


def _ptr(x, dtype=np.float64):
    if _is_torch(x):
        return C.c_void_p(x.data_ptr()), x
    a = np.ascontiguousarray(x, dtype=dtype)
    return C.c_void_p(a.ctypes.data), a


def linearise_samples(x, u, x_next, wind, lower, upper, device=0, out=None):
    """Stage linearisation of n independent samples -> (jac [n][180], res [n][12])."""
    lib = _lib.load()
    dt = np.dtype(lib.np_real)
    n = int(x.shape[0])
    px, kx = _ptr(x, dt); pu, ku = _ptr(u, dt); pn, kn = _ptr(x_next, dt); pw, kw = _ptr(wind, dt)
    if out is None:
        jac, res = np.zeros((n, 180), dtype=dt), np.zeros((n, 12), dtype=dt)
    else:
        jac, res = out
    pj, _ = (_ptr(jac) if _is_torch(jac) else (C.c_void_p(jac.ctypes.data), jac))
    pr, _ = (_ptr(res) if _is_torch(res) else (C.c_void_p(res.ctypes.data), res))
    lo, up = _vec(lower, 3, dt), _vec(upper, 3, dt)
    _lib.check(lib, lib.nmpc_linearise_samples(
        int(device), None, n, px, pu, pn, pw, lo.ctypes.data_as(lib.preal),
        up.ctypes.data_as(lib.preal), pj, pr))
    return jac, res


def rk4_samples(x, u, wind, delta=1.0 / 50.0, device=0, out=None):
    lib = _lib.load()
    dt = np.dtype(lib.np_real)
    n = int(x.shape[0])
    px, kx = _ptr(x, dt); pu, ku = _ptr(u, dt); pw, kw = _ptr(wind, dt)
    o = np.zeros((n, 13), dtype=dt) if out is None else out
    po = C.c_void_p(o.data_ptr()) if _is_torch(o) else C.c_void_p(o.ctypes.data)
    _lib.check(lib, lib.nmpc_rk4_samples(int(device), None, n, px, pu, pw, float(delta), po))
    return o
