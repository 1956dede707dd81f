"""The hypothesis property tests of test_emu_property.py on the REAL kernels, through the C-ABI: adversarial small clouds
(NaN / Inf / denormal / huge coordinates, cell and ROI boundaries, duplicates, empty frames) against the oracle."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from test_emu_property import _cloud, point

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(mot, hip_lib):
    c = mot.Context(max_points=8192, max_batch=1)
    yield c
    c.close()


@settings(max_examples=200, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(pts=st.lists(point, min_size=0, max_size=120), reps=st.integers(1, 3))
def test_ground_stage_matches_oracle(ctx, oracle, pts, reps):
    p = oracle.params(0)
    cloud = _cloud(pts, reps)
    r = ctx.ground_remove(cloud)
    g = oracle.ground_remove(p, cloud)
    assert np.array_equal(r["mask"], g["mask"])
    assert np.array_equal(r["elevated"].view(np.uint32), g["elevated"].view(np.uint32))
    assert np.array_equal(r["ground"].view(np.uint32), g["ground"].view(np.uint32))


@settings(max_examples=100, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(pts=st.lists(point, min_size=0, max_size=150), reps=st.integers(1, 40))
def test_cluster_box_side_match_oracle(ctx, oracle, pts, reps):
    p = oracle.params(0)
    elev = _cloud(pts, reps)
    r = ctx.cluster(elev)
    o = oracle.cluster(p, elev)
    assert r["num_cluster"] == o["num_cluster"] and np.array_equal(r["grid"], o["grid"])
    assert np.array_equal(r["point_label"], o["point_label"])
    b = ctx.box_fit(elev, o["grid"], o["num_cluster"])
    ob = oracle.box_fit(p, elev, o["grid"], o["num_cluster"])
    assert b["n_undefined"] == ob["n_undefined"] and np.array_equal(b["box_cluster"], ob["box_cluster"])
    assert np.array_equal(b["boxes"].view(np.uint32), ob["boxes"].view(np.uint32))
    sd = ctx.cluster_products_host(elev, o["grid"]); osd = oracle.cluster_products(p, elev, o["grid"])
    for k in ("clustered", "obstacles", "cost_map"):
        assert sd[k].shape == osd[k].shape and np.array_equal(sd[k], osd[k]), k
