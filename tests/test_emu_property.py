"""Property tests (hypothesis) of the kernels' logic under the emulator against the oracle on small adversarial clouds:
NaN / Inf / denormal / huge coordinates, points on cell and ROI boundaries, duplicates, empty and single-point frames.
(The emulator is a development aid — see tests/emu/hipemu.h; the -m gpu suite is the parity claim.)"""
import os
import sys

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))

SPECIAL = [0.0, -0.0, np.nan, np.inf, -np.inf, 1e-40, -1e-40, 3.4, -3.4, 120.0, -120.0, 25.0, -25.0, 24.999998, -24.999998,
           1e9, -1e9, 3.4028235e38, 0.2, 0.1, 8.0, -5.0, 4.5, 2.0]
SCALE = int(os.environ.get("MOT_PROP_SCALE", "1"))   # MOT_PROP_SCALE=20: a long exploration run (CPU only)
coord = st.one_of(st.sampled_from(SPECIAL), st.floats(-130, 130, width=32), st.floats(-30, 30, width=32))
zval = st.one_of(st.sampled_from([np.nan, np.inf, -np.inf, -2.0, -0.4, -1.75, 0.1, 1000.0, -99.0, 0.0, -0.0]), st.floats(-4, 3, width=32))
point = st.tuples(coord, coord, zval)


@pytest.fixture(scope="module")
def emu_ctx(mot):
    import build_emu
    c = mot.Context(lib_path=build_emu.build(), max_points=8192, max_batch=1)
    yield c
    c.close()


def _cloud(pts, reps):
    a = np.zeros((len(pts), 4), np.float32)
    if pts:
        a[:, :3] = np.array(pts, np.float32)
    return np.repeat(a, reps, axis=0) if len(a) else a


@settings(max_examples=300 * SCALE, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(pts=st.lists(point, min_size=0, max_size=120), reps=st.integers(1, 3))
def test_ground_stage_matches_oracle(emu_ctx, oracle, pts, reps):
    p = oracle.params(0)
    cloud = _cloud(pts, reps)
    r = emu_ctx.ground_remove(cloud)
    g = oracle.ground_remove(p, cloud)
    assert np.array_equal(r["mask"], g["mask"])
    assert np.array_equal(r["elevated"].view(np.uint32), g["elevated"].view(np.uint32))
    assert np.array_equal(r["ground"].view(np.uint32), g["ground"].view(np.uint32))


@settings(max_examples=150 * SCALE, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(pts=st.lists(point, min_size=0, max_size=150), reps=st.integers(1, 40))
def test_cluster_box_side_match_oracle(emu_ctx, oracle, pts, reps):
    p = oracle.params(0)
    elev = _cloud(pts, reps)   # repeats make cells pass the count > 1 rule and clusters pass the size filters
    r = emu_ctx.cluster(elev)
    o = oracle.cluster(p, elev)
    assert r["num_cluster"] == o["num_cluster"] and np.array_equal(r["grid"], o["grid"])
    assert np.array_equal(r["point_label"], o["point_label"])
    b = emu_ctx.box_fit(elev, o["grid"], o["num_cluster"])
    ob = oracle.box_fit(p, elev, o["grid"], o["num_cluster"])
    assert b["n_undefined"] == ob["n_undefined"] and np.array_equal(b["box_cluster"], ob["box_cluster"])
    assert np.array_equal(b["boxes"].view(np.uint32), ob["boxes"].view(np.uint32))
    sd = emu_ctx.cluster_products_host(elev, o["grid"]); osd = oracle.cluster_products(p, elev, o["grid"])
    for k in ("clustered", "obstacles", "cost_map"):
        assert sd[k].shape == osd[k].shape and np.array_equal(sd[k], osd[k]), k
