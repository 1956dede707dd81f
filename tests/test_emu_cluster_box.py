"""CPU-only development check of the clustering / box kernels' LOGIC under tests/emu/hipemu.h (see that header:
not a product path, not a parity claim)."""
import os
import sys

import numpy as np
import pytest

import patterns

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.fixture(scope="module")
def emu_lib():
    import build_emu
    return build_emu.build()


def test_emu_cluster_box_pipeline(mot, emu_lib, oracle, synth):
    p = oracle.params(0)
    with mot.Context(lib_path=emu_lib, max_points=40000) as c:
        for stream in (1, 2):
            cloud = synth.make_cloud(36000, stream, 0)
            g = c.ground_remove(cloud)
            og = oracle.ground_remove(p, cloud)
            assert np.array_equal(g["elevated"], og["elevated"])
            r = c.cluster(g["elevated"]); o = oracle.cluster(p, og["elevated"])
            assert r["num_cluster"] == o["num_cluster"] and np.array_equal(r["grid"], o["grid"]) and np.array_equal(r["point_label"], o["point_label"])
            b = c.box_fit(g["elevated"], r["grid"], r["num_cluster"]); ob = oracle.box_fit(p, og["elevated"], o["grid"], o["num_cluster"])
            assert np.array_equal(b["boxes"], ob["boxes"]) and np.array_equal(b["box_cluster"], ob["box_cluster"])


def test_emu_golden_frame_ot0(mot, emu_lib):
    """the KITTI-tuned preset end to end under the emulator against the fixture from object_tracking0's own sources"""
    import golden_util as G
    fx = G.load(G.FRAMES_OT0[0])
    with mot.Context(mot.params(1, lib=mot.load_library(emu_lib)), lib_path=emu_lib, max_points=65536) as c:
        g = c.ground_remove(fx["cloud"])
        cl = c.cluster(g["elevated"])
        bx = c.box_fit(g["elevated"], cl["grid"], cl["num_cluster"])
        G.check_frame(fx, g, cl, bx["boxes"])


def test_emu_side_products(mot, emu_lib, oracle, synth):
    """makeClusteredCloud / setObsMsg / createCostMap kernel under the emulator against the restatement"""
    p = oracle.params(0)
    with mot.Context(lib_path=emu_lib, max_points=40000) as c:
        elev = np.concatenate([oracle.ground_remove(p, synth.make_cloud(36000, 2, 1))["elevated"], synth.edge_case_points()])
        r = c.cluster(elev)
        a = c.cluster_products(0); o = oracle.cluster_products(p, elev, r["grid"])
        assert len(o["clustered"]) > 100 and len(o["obstacles"]) > 10
        for k in ("clustered", "obstacles", "cost_map"):
            assert a[k].shape == o[k].shape and np.array_equal(a[k], o[k]), k
        h = c.cluster_products_host(elev[:5000], r["grid"]); oh = oracle.cluster_products(p, elev[:5000], r["grid"])
        for k in ("clustered", "obstacles", "cost_map"):
            assert np.array_equal(h[k], oh[k]), k


def test_emu_box_markers(mot, emu_lib, oracle, synth):
    """mot_box_markers (the rviz cubes: mark_cluster, box_fitting.cpp:161-209) under the emulator: equal to the float32 restatement in
    numpy bit for bit, and to the markers the reference's own boxFitting fills; stage-wise and fused slots"""
    import oracle_lib as O
    p = oracle.params(0)
    with mot.Context(lib_path=emu_lib, max_points=40000, max_batch=2) as c:
        clouds = [synth.make_cloud(36000, 2, 1), synth.make_cloud(30000, 5, 0)]
        elev = oracle.ground_remove(p, clouds[0])["elevated"]
        r = c.cluster(elev)
        b = c.box_fit_resident()
        m = c.box_markers(0)
        assert len(m) == len(b["boxes"]) > 5
        want = O.box_markers_numpy(elev, r["point_label"], b["box_cluster"])
        assert np.array_equal(m.view(np.uint32), want.view(np.uint32))
        if O.ref() is not None:
            rm = O.ref_box_markers(elev, r["grid"], r["num_cluster"])
            mine = m.astype(np.float64); mine[:, 3:][mine[:, 3:] == 0] = 0.1
            assert np.array_equal(mine, rm)
        with pytest.raises(mot.MotError):
            c.box_markers(0, max_boxes=len(m) - 1)
        # every slot of a fused batch
        host = np.zeros((2, 40000, 4), np.float32)
        for s, cl in enumerate(clouds):
            host[s, : len(cl)] = cl
        c.frames_dev(host.ctypes.data, 40000 * 4, [len(cl) for cl in clouds])
        for s, cl in enumerate(clouds):
            e = oracle.ground_remove(p, cl)["elevated"]
            k = c.get_clusters(s, len(e)); bb = c.get_boxes(s)
            assert np.array_equal(c.box_markers(s).view(np.uint32), O.box_markers_numpy(e, k["point_label"], bb["box_cluster"]).view(np.uint32))
        # no boxes: nothing to mark
        c.cluster(np.zeros((0, 4), np.float32)); c.box_fit_resident()
        assert c.box_markers(0).shape == (0, 6)


@pytest.mark.parametrize("preset", [0, 1])
def test_emu_ccl_patterns(mot, emu_lib, oracle, preset):
    rng = np.random.default_rng(1)
    p = oracle.params(preset)
    with mot.Context(mot.params(preset, lib=mot.load_library(emu_lib)), lib_path=emu_lib, max_points=150000) as c:
        for name, cells in patterns.occupancy_cases(p.num_grid, rng, dense=False):
            pts = patterns.case_points(cells, p, rng)
            r = c.cluster(pts); o = oracle.cluster(p, pts)
            assert r["num_cluster"] == o["num_cluster"] and np.array_equal(r["grid"], o["grid"]), name
            assert np.array_equal(r["point_label"], o["point_label"]), name


def test_emu_many_clusters_per_tile(mot, emu_lib, oracle):
    from test_cluster_box_gpu import interleaved_clusters_cloud
    p = oracle.params(0)
    with mot.Context(lib_path=emu_lib, max_points=4096) as c:
        for cloud in (interleaved_clusters_cloud(), interleaved_clusters_cloud(20, 25)):
            r = c.cluster(cloud); o = oracle.cluster(p, cloud)
            assert r["num_cluster"] == o["num_cluster"] >= 10 and np.array_equal(r["grid"], o["grid"])
            b = c.box_fit(cloud, o["grid"], o["num_cluster"]); ob = oracle.box_fit(p, cloud, o["grid"], o["num_cluster"])
            assert np.array_equal(b["boxes"], ob["boxes"]) and np.array_equal(b["box_cluster"], ob["box_cluster"])


def test_emu_shuffled_many_clusters(mot, emu_lib, oracle):
    from test_cluster_box_gpu import shuffled_many_clusters_cloud
    p = oracle.params(0)
    cloud = shuffled_many_clusters_cloud(40, 300)
    with mot.Context(lib_path=emu_lib, max_points=32768) as c:
        r = c.cluster(cloud); o = oracle.cluster(p, cloud)
        assert r["num_cluster"] == o["num_cluster"] and np.array_equal(r["grid"], o["grid"])
        b = c.box_fit(cloud, o["grid"], o["num_cluster"]); ob = oracle.box_fit(p, cloud, o["grid"], o["num_cluster"])
        assert np.array_equal(b["boxes"], ob["boxes"]) and np.array_equal(b["box_cluster"], ob["box_cluster"])
        # the cubes of a frame whose clusters are interleaved point by point (hundreds of one-point groups per cluster, the index
        # kernel's general path): still the sums in input order
        import oracle_lib as O
        assert len(b["boxes"]) > 0
        assert np.array_equal(c.box_markers(0).view(np.uint32), O.box_markers_numpy(cloud, o["point_label"], b["box_cluster"]).view(np.uint32))

ZERO_HEIGHT_CASES = ([np.nan, np.nan, np.nan, -0.0, np.nan, np.nan], [np.nan, 0.0, np.nan, -0.0, np.nan, np.nan], [-1.0, -0.0, 0.0, -0.5, np.nan, np.nan],
                     [-1.0, -2.0, -0.0, -0.0, 0.0, np.nan], [0.5, -0.0, 0.0, np.nan, np.nan, np.nan], [-0.0] * 6, [0.0] + [-0.0] * 5)


def _zero_height_cloud(zs):
    pts = [(0.0, 0.0, z) for z in zs[:4]] + [(0.0, 0.5, zs[4]), (0.5, 0.0, zs[5])]
    a = np.zeros((len(pts), 4), np.float32); a[:, :3] = np.array(pts, np.float32)
    return np.repeat(a, 5, axis=0)


def test_emu_box_height_sign_of_zero(mot, emu_lib, oracle):
    """`if (pZ > maxZ) maxZ = pZ` keeps the first of equal maxima; -0 and +0 are equal: the box's top face must carry the sign
    of the FIRST zero of the cluster (found by the hypothesis tests: the keyed maximum alone returned +0)"""
    p = oracle.params(0)
    with mot.Context(lib_path=emu_lib, max_points=4096) as c:
        for zs in ZERO_HEIGHT_CASES:
            e = _zero_height_cloud(zs)
            o = oracle.cluster(p, e)
            b = c.box_fit(e, o["grid"], o["num_cluster"]); ob = oracle.box_fit(p, e, o["grid"], o["num_cluster"])
            assert len(ob["boxes"]) == 1 and np.array_equal(b["boxes"].view(np.uint32), ob["boxes"].view(np.uint32)), zs


def test_emu_wide_clusters_take_the_large_hull_kernel(mot, emu_lib, oracle):
    """frames with and without clusters of more than 510 candidate hull points, alternating on one context: the per-frame flag the
    gather kernel leaves for cluster_rect_large_kernel must be raised, honoured and re-armed"""
    import wide_clusters as W
    p = oracle.params(0, t_len_max=100.0, t_area_max=200.0, t_width_max=10.0, t_ratio_max=500.0, t_pt_per_m3=0.1)   # let the long boxes through the rule filter
    mp = mot.params(0, lib=mot.load_library(emu_lib), t_len_max=100.0, t_area_max=200.0, t_width_max=10.0, t_ratio_max=500.0, t_pt_per_m3=0.1)
    with mot.Context(mp, lib_path=emu_lib, max_points=32768) as c:
        wide = 0
        for seed, walls in ((0, 2), (1, 0), (2, 1), (3, 0), (4, 2)):
            bx, got = W.check(c, oracle, p, W.wide_wall_cloud(seed, walls))
            wide += sum(1 for d in bx["debug"] if d["branch"] == 1 and d["num_points"] > 5000 and d["accepted"])   # a box that only the large-hull kernel can have produced
        assert wide >= 4


def test_emu_l_shape_cluster_with_more_groups_than_the_staging_holds(mot, emu_lib, oracle):
    """the L-shape branch's "r-th point of the cluster" search over the cluster's groups, with more groups than are staged in LDS (the
    emulator build stages 48) and more than one batch of records per wave (8 there): a dense car-sized L in random order, both RNG mappings"""
    import wide_clusters as W
    for mapping in (1, 0):
        kw = dict(rng_mapping=mapping)
        p = oracle.params(0, **kw)
        with mot.Context(mot.params(0, lib=mot.load_library(emu_lib), **kw), lib_path=emu_lib, max_points=16384) as c:
            bx, got = W.check(c, oracle, p, W.big_l_cloud(1, n_per_wall=5000))
            assert any(d["branch"] == 0 and d["num_points"] >= 8000 and d["accepted"] for d in bx["debug"])
