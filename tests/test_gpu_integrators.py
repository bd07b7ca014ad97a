"""GPU tests (pytest -m gpu) of libcnmpc_heun.so and libcnmpc_euler.so: the
library built with the Heun / Euler integrator of the reference's compile-time
matrix (include/config.h:30-39, include/integrator.h:66-91).  The reference has
these integrators only in its C++ variant, which cannot be built offline, so no
reference output exists for them: PARITY UNPINNED with respect to the reference.
The checks are bit-for-bit against the CPU port built the same way (whose
integrators pass an order-of-accuracy test on the CPU)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r); sys.path.insert(0, %(oracle)r)
from helpers import assert_same, same_bits
from nmpc_amd import _lib
from nmpc_amd import synthetic as S
from nmpc_amd.batch import Batch, linearise_samples, rk4_samples
import portdriver as P
which = %(which)r
assert _lib.LIB_PATH.endswith("libcnmpc_%%s.so" %% which)

rng = np.random.default_rng(20131118)
x, u, w = S.random_states(2048, rng)
got = rk4_samples(x, u, w)
exp = np.stack([P.rk4(x[i], u[i], w[i], precision=which) for i in range(len(x))])
assert_same(got, exp, "one integration step")
assert not np.array_equal(got, np.stack([P.rk4(x[i], u[i], w[i]) for i in range(64)] + list(got[64:])))

xn = np.roll(x, 3, axis=0)[:512]
lower, upper = np.array([0.0, -0.1, 0.05]), np.array([1.0, 0.9, 1.3])
jac, res = linearise_samples(x[:512], u[:512], xn, w[:512], lower, upper)
ejac, eres, _ = P.batch_linearise(x[:512], u[:512], xn, w[:512], lower, upper, threads=4, precision=which)
assert_same(jac, ejac, "Jacobian"); assert_same(res, eres, "residual")

N, B, steps = 60, 96, 3
d = S.make_batch(B, N, extra_steps=steps, seed=31)
cfg = S.setup_args()
bt = Batch(B, N); bt.setup(**cfg); bt.initialise_horizon(); bt.set_wind_velocity(d["wind"])
bt.set_reference(d["refs"][:, :N + 1])
ports = []
for b in range(B):
    p = P.Port(N, which)
    p.setup(cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"],
            cfg["upper_control_bound"], cfg["lower_control_bound"])
    p.init(); p.set_wind(d["wind"][b])
    for i in range(N + 1):
        p.set_reference(d["refs"][b, i], i)
    ports.append(p)
good = 0
for k in range(steps):
    bt.solve(d["meas"][:, k])
    uu, st, it = bt.get_controls()
    for b, p in enumerate(ports):
        p.feedback(d["meas"][b, k])
        eu, es = p.controls()
        assert same_bits(uu[b], eu).all() and st[b] == es and it[b] == p.num_iter(), (k, b)
        p.update_horizon(d["refs"][b, N + 1 + k])
    good += int((st == 0).sum())
    bt.update_horizon(d["refs"][:, N + 1 + k])
assert good > B
bt.close()
print("child ok")
"""


@pytest.mark.parametrize("which", ["heun", "euler"])
def test_integrator_build_matches_port(port_built, which):
    lib = os.path.join(ROOT, "nmpc_amd", "libcnmpc_%s.so" % which)
    assert os.path.exists(lib), "build it with `make -C nmpc_amd/csrc`"
    code = CHILD % dict(root=ROOT, tests=os.path.join(ROOT, "tests"), oracle=os.path.join(ROOT, "oracle"),
                        which=which)
    env = dict(os.environ, NMPC_AMD_LIB=lib)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
