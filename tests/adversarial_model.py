"""Model and stage linearisation on hostile inputs (huge, tiny, zero, inf, NaN entries) against the port, by bit pattern."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import portdriver as P
from nmpc_amd import synthetic as S
from nmpc_amd.batch import linearise_samples, rk4_samples
from helpers import same_bits
PREC = os.environ.get("SOAK_PRECISION", "double")
dt = np.float32 if PREC == "float" else np.float64
n = int(os.environ.get("COUNT", "20000"))
rng = np.random.default_rng(int(os.environ.get("SEED", "7")))
x, u, w = S.random_states(n, rng, spread=float(os.environ.get("SPREAD", "3")))
xn = np.roll(x, 1, axis=0).copy()
special = np.array([0.0, -0.0, 1e-310, -1e-300, 1e150, -1e200, 1e308, np.inf, -np.inf, np.nan, 1.0, -1.0, 1e-20, 3e38, 1e-45])
hit = rng.random(n) < 0.5
for arr in (x, u, w, xn):
    m = (rng.random(arr.shape) < 0.08) & hit[:, None]
    arr[m] = special[rng.integers(0, len(special), int(m.sum()))]
x, u, w, xn = (np.ascontiguousarray(a.astype(dt)) for a in (x, u, w, xn))
lo, up = np.array([0.0, 0.0, 0.0], dt), np.array([1.0, 1.0, 1.0], dt)
with np.errstate(all="ignore"):
    xp = rk4_samples(x, u, w)
    jac, res = linearise_samples(x, u, xn, w, lo, up)
def one(i):
    return P.rk4(x[i], u[i], w[i], precision=PREC), P.linearise(x[i], u[i], xn[i], w[i], lo, up, precision=PREC)
with ThreadPoolExecutor(16) as pool:
    exp = list(pool.map(one, range(n)))
bad_r = [i for i in range(n) if not same_bits(xp[i], np.asarray(exp[i][0], dt)).all()]
bad_j = [i for i in range(n) if not (same_bits(jac[i].ravel(), np.asarray(exp[i][1][0], dt).ravel()).all() and same_bits(res[i], np.asarray(exp[i][1][1], dt)).all())]
nonfinite = int((~np.isfinite(xp)).any(axis=1).sum())
print(PREC, "samples", n, "with a non-finite successor", nonfinite, "| rk4 mismatches", len(bad_r), bad_r[:5], "| linearisation mismatches", len(bad_j), bad_j[:5])
for i in (bad_r[:2] + bad_j[:2]):
    print(" sample", i, "x", x[i].tolist(), "u", u[i].tolist(), "w", w[i].tolist())
    print("  gpu x+", xp[i].tolist()); print("  cpu x+", np.asarray(exp[i][0]).tolist())
    d = ~same_bits(jac[i].ravel(), np.asarray(exp[i][1][0], dt).ravel())
    print("  jac diffs at", np.argwhere(d)[:8, 0].tolist(), jac[i].ravel()[d][:4].tolist(), np.asarray(exp[i][1][0], dt).ravel()[d][:4].tolist())
