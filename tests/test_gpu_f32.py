"""GPU parity tests (pytest -m gpu) of libcnmpc_f32.so, the library built in
the reference's shipped single-precision configuration (real_t = float):
against golden vectors generated from the reference compiled in that
configuration (tests/golden/*_f32.npz) and against the single-precision CPU
oracle port on seeded inputs.  Bit-for-bit, like the double-precision tests.
The library is chosen per process (NMPC_AMD_LIB), so the checks run in a child."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
from helpers import SEQUENCES_F32, assert_same, assert_same_state, same_bits, drive_gpu_batch, golden
from nmpc_amd import _lib
from nmpc_amd.batch import Batch, linearise_samples, rk4_samples
lib = _lib.load()
assert _lib.LIB_PATH.endswith("libcnmpc_f32.so") and lib.np_real == "float32"
assert lib.nmpc_config_get_precision() == 0          # NMPC_PRECISION_FLOAT

g = golden("model_f32")
assert_same(rk4_samples(g["x"], g["u"], g["wind"]), g["xplus"], "x+ (RK4)")
g = golden("linearise_f32")
jac, res = linearise_samples(g["x"], g["u"], g["xnext"], g["wind"], g["lower"], g["upper"])
assert jac.dtype == np.float32
assert_same(jac, g["jac"], "Jacobian")
assert_same(res, g["res"], "residual")

for name in SEQUENCES_F32:
    g = golden(name)
    dump = "final_stages" in g
    out = drive_gpu_batch([g], dump=dump)
    assert_same(out["status"][0], g["status"], name + " status")
    assert_same(out["iters"][0], g["iters"], name + " QP iterations")
    assert_same(out["u"][0], g["u"], name + " controls")
    assert_same(out["z0"][0], g["z0"], name + " z0")
    if dump:
        assert_same(out["loaded_Cc"][0], g["loaded_Cc"], name + " C,c after loading")
        assert_same(out["final_stages"][0], g["final_stages"], name + " final stage records")
        assert_same_state(out["final_globals"][0], g["final_globals"], name + " final solver state")
# the four turn sequences as one batch
gs = [golden("seq_turn%%d_f32" %% b) for b in range(4)]
out = drive_gpu_batch(gs)
for b, g in enumerate(gs):
    assert_same(out["u"][b], g["u"], "batched turn%%d controls" %% b)
    assert_same(out["iters"][b], g["iters"], "batched turn%%d iterations" %% b)

# seeded batches, cold solve and warm steps, against the single-precision port: the
# default horizon, and one beyond the range of the 1024-lane step-length search kernel
import portdriver as P
from nmpc_amd import synthetic as S
for N, B, steps, seed in ((100, 300, 3, 808), (150, 24, 2, 809)):
    d = S.make_batch(B, N, extra_steps=steps, seed=seed)
    cfg = S.setup_args()
    bt = Batch(B, N); bt.setup(**cfg); bt.initialise_horizon(); bt.set_wind_velocity(d["wind"])
    bt.set_reference(d["refs"][:, :N + 1])
    ports = []
    for b in range(B):
        p = P.Port(N, "float")
        p.setup(cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"],
                cfg["upper_control_bound"], cfg["lower_control_bound"])
        p.init(); p.set_wind(d["wind"][b])
        for i in range(N + 1):
            p.set_reference(d["refs"][b, i], i)
        ports.append(p)
    converged = 0
    for k in range(steps):
        bt.solve(d["meas"][:, k])
        u, st, it = bt.get_controls()
        for b, p in enumerate(ports):
            p.feedback(d["meas"][b, k])
            eu, es = p.controls()
            assert same_bits(u[b], eu).all() and st[b] == es and it[b] == p.num_iter(), (N, k, b, u[b], eu, st[b], es, it[b], p.num_iter())
            p.update_horizon(d["refs"][b, N + 1 + k])
        converged += int((st == 0).sum())
        bt.update_horizon(d["refs"][:, N + 1 + k])
    assert converged > B // 2, (N, converged)
    bt.close()

# A step-length search that fails (no ascent direction: the solve ends there,
# dual_qp.c:1294-1334) must leave the solver state the reference leaves, the
# gradient record of the iteration's set-up included: instance 195 of this
# seeded batch ends its sixth step that way.
N, steps, first, count = 100, 6, 192, 8
d = S.make_batch(4096, N, extra_steps=steps, seed=108)
d = {k: v[first:first + count] for k, v in d.items()}
cfg = S.setup_args()
bt = Batch(count, N); bt.setup(**cfg); bt.initialise_horizon(); bt.set_wind_velocity(d["wind"])
bt.set_reference(d["refs"][:, :N + 1])
for k in range(steps):
    bt.solve(d["meas"][:, k]); u, st, it = bt.get_controls()
    if k + 1 < steps:
        bt.update_horizon(d["refs"][:, N + 1 + k])
assert bt.solver_info(195 - first)["status"] == 13 and it[195 - first] < 5, bt.solver_info(195 - first)
for b in range(count):
    p = P.Port(N, "float")
    p.setup(cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"],
            cfg["upper_control_bound"], cfg["lower_control_bound"])
    p.init(); p.set_wind(d["wind"][b])
    for i in range(N + 1):
        p.set_reference(d["refs"][b, i], i)
    for k in range(steps):
        p.feedback(d["meas"][b, k])
        if k + 1 < steps:
            p.update_horizon(d["refs"][b, N + 1 + k])
    eu, es = p.controls()
    assert same_bits(u[b], eu).all() and st[b] == es and it[b] == p.num_iter(), b
    assert_same(bt.stages(b), p.stages(), "stage records, instance %%d" %% (first + b))
    assert_same_state(bt.globals_(b), p.globals_(), "solver state, instance %%d" %% (first + b))
bt.close()
print("child ok")
"""


def test_single_precision_library(port_built):
    lib = os.path.join(ROOT, "nmpc_amd", "libcnmpc_f32.so")
    assert os.path.exists(lib), "build it with `make -C nmpc_amd/csrc`"
    code = CHILD % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"))
    env = dict(os.environ, NMPC_AMD_LIB=lib)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
