"""CPU tests of host-side logic: synthetic input generator and instance
partitioning."""
import numpy as np

from nmpc_amd import sharding
from nmpc_amd import synthetic as S


def test_make_batch_shapes_and_determinism():
    a = S.make_batch(5, 20, extra_steps=3, seed=1)
    b = S.make_batch(5, 20, extra_steps=3, seed=1)
    assert a["refs"].shape == (5, 24, 16) and a["meas"].shape == (5, 4, 13)
    assert a["wind"].shape == (5, 3)
    for k in a:
        assert np.array_equal(a[k], b[k])
    c = S.make_batch(5, 20, extra_steps=3, seed=2)
    assert not np.array_equal(a["refs"], c["refs"])


def test_reference_points_are_physical():
    d = S.make_batch(64, 100, extra_steps=10, seed=3, turn_start=(0.3, 3.0))
    r = d["refs"]
    q = r[:, :, 6:10]
    np.testing.assert_allclose(np.linalg.norm(q, axis=-1), 1.0, atol=1e-12)
    assert np.all(r[:, :, 13:] == 0.5)
    speed = np.linalg.norm(r[:, 0, 3:6], axis=-1)
    assert speed.min() >= 15 and speed.max() <= 25
    # level flight: body x axis (rotated into NED by the conjugate) along velocity
    qq = r[:, 0, 6:10]
    psi = -2 * np.arctan2(qq[:, 2], qq[:, 3])
    v = r[:, 0, 3:6]
    np.testing.assert_allclose(np.cos(psi) * speed, v[:, 0], atol=1e-9)
    np.testing.assert_allclose(np.sin(psi) * speed, v[:, 1], atol=1e-9)
    # positions advance by velocity * dt on the first leg
    step = r[:, 1, 0:3] - r[:, 0, 0:3]
    np.testing.assert_allclose(step, v * S.STEP_LENGTH, atol=1e-9)


def test_straight_reference_matches_check_scenario():
    r = S.straight_reference(5)
    assert np.array_equal(r[3], [1.2000000000000002, 0, 0, 20, 0, 0, 0, 0, 0, 1, 0, 0, 0, .5, .5, .5])
    m = S.straight_measurement(r[2])
    assert m[1] == 0.5 and m[5] == 0.2 and m[0] == r[2, 0]


def test_partition_is_contiguous_and_complete():
    for total in (1, 7, 4096, 262144):
        for world in (1, 2, 3, 8):
            spans = [sharding.instance_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_state_plus_delta_inverts_state_to_delta(port_built):
    """x (+) d of the real-time-iteration preparation step is the inverse of the
    reference's state difference (ccs-c66x/cnmpc.c:541-579) up to rounding."""
    import ctypes as C
    import portdriver as P
    from nmpc_amd import synthetic as S
    lib = P.lib()
    dp = C.POINTER(C.c_double)
    lib.port_state_plus_delta.argtypes = [dp, dp, dp]
    rng = np.random.default_rng(4)
    x, _, _ = S.random_states(200, rng)
    for i in range(200):
        d = rng.normal(0, 0.3, 12)
        out = np.zeros(13)
        lib.port_state_plus_delta(out.ctypes.data_as(dp), x[i].ctypes.data_as(dp), d.ctypes.data_as(dp))
        back = P.state_to_delta(x[i], out)
        np.testing.assert_allclose(back, d, rtol=0, atol=1e-12)
    z = np.zeros(12)
    out = np.zeros(13)
    lib.port_state_plus_delta(out.ctypes.data_as(dp), x[0].ctypes.data_as(dp), z.ctypes.data_as(dp))
    assert np.array_equal(out, x[0])


def test_factor_block_row_layout_is_a_packing():
    """The factor's block row (nmpc_amd/csrc/layout.hpp): the 12x12 sub-diagonal
    block, then the lower triangle of the diagonal block.  Rows of the triangle
    start at even offsets (16-byte pieces), do not overlap each other or the
    sub-diagonal block, stay inside the row, and none crosses a multiple of 24
    values (one LDS-DMA piece of the Newton solve)."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(__file__), "..", "nmpc_amd", "csrc", "layout.hpp")).read()
    row = int(re.search(r"kCholRow\s*=\s*(\d+)", src).group(1))
    body = re.search(r"kCholDiagRow\[12\]\s*=\s*\{([^}]*)\}", src).group(1)
    starts = [sum(int(x) for x in e.split("+")) for e in body.split(",")]         # entries are sums of two literals
    slack = int(re.search(r"kCholSlack\s*=\s*(\d+)", src).group(1))
    assert len(starts) == 12
    used = np.zeros(row, dtype=int)
    used[:144] += 1                                     # rows of twelve: the sub-diagonal block
    for l, at in enumerate(starts):
        assert at % 2 == 0
        assert at // 24 == (at + l) // 24               # entries 0..l in one piece
        used[at:at + l + 1] += 1
        # a lane reads the first twelve values of any row: inside the row plus the slack
        assert at + 12 <= row + slack
    assert used.max() == 1
    assert used.sum() == 144 + 78
