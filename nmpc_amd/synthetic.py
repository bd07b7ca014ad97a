"""Synthetic NMPC inputs: per-instance reference trajectories, measurements and
wind, shaped like the reference's closed-loop harness builds them
(scripts/nmpc-xplane-synthetic.py:56-78 piecewise-linear waypoint interpolation
with quaternion normalisation, :87-92 weights and bounds, :97-143 a straight
leg followed by a 90 degree turn, :193-197 horizon loading with reference
controls 0.5).  numpy only; no device code."""
import numpy as np

STATE_WEIGHTS = [1, 1, 1, 1, 1, 1, 1, 1, 1e1, 7e-1, 7e-1, 1e1]
CONTROL_WEIGHTS = [1e-1, 1e3, 1e3]
TERMINAL_WEIGHTS = [1.0] * 12
UPPER_CONTROL_BOUND = [1.0, 1.0, 1.0]
LOWER_CONTROL_BOUND = [0.0, 0.0, 0.0]
STEP_LENGTH = 1.0 / 50.0
DEFAULT_SEED = 20131118


def setup_args():
    return dict(state_weights=STATE_WEIGHTS, control_weights=CONTROL_WEIGHTS,
                terminal_weights=TERMINAL_WEIGHTS,
                upper_control_bound=UPPER_CONTROL_BOUND,
                lower_control_bound=LOWER_CONTROL_BOUND)


def yaw_quaternion(psi):
    """Attitude (x, y, z, w) of level flight on heading psi: the harness's
    euler_to_q(yaw, 0, 0) = rotation about Z by -yaw."""
    psi = np.asarray(psi, dtype=np.float64)
    z = np.zeros_like(psi)
    return np.stack([z, z, np.sin(-0.5 * psi), np.cos(-0.5 * psi)], axis=-1)


def straight_reference(n_points, speed=20.0, step=0.4):
    """The single-instance check scenario: level flight north at `speed`,
    reference points `step` metres apart, identity attitude, controls 0.5."""
    i = np.arange(n_points, dtype=np.float64)
    ref = np.zeros((n_points, 16))
    ref[:, 0] = step * i
    ref[:, 3] = speed
    ref[:, 9] = 1.0
    ref[:, 13:] = 0.5
    return ref


def straight_measurement(ref_point, east=0.5, sink=0.2):
    m = np.array(ref_point[:13], dtype=np.float64)
    m[1] = east
    m[5] = sink
    return m


def waypoint_batch(batch, rng, turn_start=(0.3, 30.0), turn_time=2.0):
    """[B][4][14] waypoints (t, pos3, vel3, quat4, omega3): straight leg, a
    quarter turn flown in `turn_time` seconds (the harness's 29 s -> 31 s
    segment), straight leg on the new heading."""
    speed = rng.uniform(15.0, 25.0, batch)
    psi = rng.uniform(0.0, 2.0 * np.pi, batch)
    t1 = rng.uniform(turn_start[0], turn_start[1], batch)
    sign = np.where(rng.random(batch) < 0.5, -1.0, 1.0)
    psi2 = psi + sign * 0.5 * np.pi
    d1 = np.stack([np.cos(psi), np.sin(psi), np.zeros(batch)], -1)
    d2 = np.stack([np.cos(psi2), np.sin(psi2), np.zeros(batch)], -1)
    v1, v2 = speed[:, None] * d1, speed[:, None] * d2
    wp = np.zeros((batch, 4, 14))
    p1 = v1 * t1[:, None]
    p2 = p1 + 0.5 * turn_time * (v1 + v2)
    t_end = t1 + turn_time + 60.0
    p3 = p2 + v2 * 60.0
    for j, (t, p, v, ps) in enumerate((
            (np.zeros(batch), np.zeros((batch, 3)), v1, psi),
            (t1, p1, v1, psi), (t1 + turn_time, p2, v2, psi2),
            (t_end, p3, v2, psi2))):
        wp[:, j, 0] = t
        wp[:, j, 1:4] = p
        wp[:, j, 4:7] = v
        wp[:, j, 7:11] = yaw_quaternion(ps)
    return wp


def interpolate(wp, times):
    """Piecewise-linear interpolation of waypoints [B][W][14] at times [M]
    -> reference points [B][M][16] (state 13, controls 0.5)."""
    B, W, _ = wp.shape
    t = np.asarray(times, dtype=np.float64)[None, :]
    out = np.zeros((B, t.shape[1], 16))
    vals = np.broadcast_to(wp[:, -1:, 1:], (B, t.shape[1], 13)).copy()
    for s in range(W - 2, -1, -1):
        t0, t1 = wp[:, s, 0:1], wp[:, s + 1, 0:1]
        frac = (t - t0) / (t1 - t0)
        seg = wp[:, s, None, 1:] + frac[:, :, None] * (wp[:, s + 1, None, 1:] - wp[:, s, None, 1:])
        mask = (t < t1)
        vals = np.where(mask[:, :, None], seg, vals)
    q = vals[:, :, 6:10]
    vals[:, :, 6:10] = q / np.linalg.norm(q, axis=-1, keepdims=True)
    out[:, :, :13] = vals
    out[:, :, 13:] = 0.5
    return out


def make_batch(batch, horizon, extra_steps=0, seed=DEFAULT_SEED, **kw):
    """Inputs for `batch` independent controllers.

    Returns dict with
      refs  [B][horizon+1+extra_steps][16]  reference points at 50 Hz
      meas  [B][1+extra_steps][13]          measurement for each control step
      wind  [B][3]                          wind estimate (N(0, 3 m/s) per axis)
    Measurements are the matching reference state plus N(0,0.5 m) position,
    N(0,0.2 m/s) velocity, a small-angle attitude error N(0,0.02 rad) and
    N(0,0.02 rad/s) body rates.
    """
    rng = np.random.default_rng(seed)
    wp = waypoint_batch(batch, rng, **kw)
    m = horizon + 1 + extra_steps
    refs = interpolate(wp, np.arange(m) * STEP_LENGTH)
    wind = rng.normal(0.0, 3.0, (batch, 3))
    base = refs[:, :1 + extra_steps, :13].copy()
    meas = base.copy()
    meas[:, :, 0:3] += rng.normal(0.0, 0.5, (batch, 1 + extra_steps, 3))
    meas[:, :, 3:6] += rng.normal(0.0, 0.2, (batch, 1 + extra_steps, 3))
    ang = rng.normal(0.0, 0.02, (batch, 1 + extra_steps, 3))
    dq = np.concatenate([0.5 * ang, np.ones((batch, 1 + extra_steps, 1))], -1)
    dq /= np.linalg.norm(dq, axis=-1, keepdims=True)
    meas[:, :, 6:10] = quat_mul(dq, base[:, :, 6:10])
    meas[:, :, 10:13] += rng.normal(0.0, 0.02, (batch, 1 + extra_steps, 3))
    return dict(refs=np.ascontiguousarray(refs), meas=np.ascontiguousarray(meas),
                wind=np.ascontiguousarray(wind))


def quat_mul(a, b):
    """Hamilton product, xyzw storage (ccs-c66x/cnmpc.c:175-178)."""
    ax, ay, az, aw = (a[..., i] for i in range(4))
    bx, by, bz, bw = (b[..., i] for i in range(4))
    return np.stack([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
        aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def random_states(count, rng, spread=1.0):
    """(x[13], u[3], wind[3]) samples that reach every branch of the aircraft
    model: stall blends on both sides, thrust clamp, the unit floors on airspeed
    and a few exactly-zero airflow components."""
    x = np.zeros((count, 13))
    x[:, 0:3] = rng.normal(0, 100, (count, 3))
    x[:, 3:6] = rng.normal(0, 12 * spread, (count, 3))
    q = rng.normal(0, 1, (count, 4))
    x[:, 6:10] = q / np.linalg.norm(q, axis=1, keepdims=True)
    x[:, 10:13] = rng.normal(0, 0.5 * spread, (count, 3))
    u = rng.uniform(-0.2, 1.2, (count, 3))
    wind = rng.normal(0, 5 * spread, (count, 3))
    slow = rng.random(count) < 0.1          # |airflow| < 1: floors engage
    x[slow, 3:6] = wind[slow] + rng.normal(0, 0.2, (slow.sum(), 3))
    still = rng.random(count) < 0.02        # exactly zero airflow
    x[still, 3:6] = wind[still]
    level = rng.random(count) < 0.3         # near-level forward flight
    n = level.sum()
    psi = rng.uniform(0, 2 * np.pi, n)
    x[level, 6:10] = yaw_quaternion(psi)
    sp = rng.uniform(5, 30, n)
    x[level, 3] = sp * np.cos(psi)
    x[level, 4] = sp * np.sin(psi)
    x[level, 5] = rng.normal(0, 3, n)
    wind[level] = rng.normal(0, 2, (n, 3))
    return x, u, wind
