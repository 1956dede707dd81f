"""Parity of the HIP clustering and box-fitting stages (through the C-ABI) against the oracle and the golden
vectors. Bit-exact: label grid, cluster count, per-point labels, box corners, box order."""
import numpy as np
import pytest

import golden_util as G
import patterns

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(mot, hip_lib):
    c = mot.Context(max_points=262144, max_batch=8)
    yield c
    c.close()


def _stage_parity(ctx, oracle, p, elev):
    r = ctx.cluster(elev)
    o = oracle.cluster(p, elev)
    assert r["num_cluster"] == o["num_cluster"]
    assert np.array_equal(r["grid"], o["grid"])
    assert np.array_equal(r["point_label"], o["point_label"])
    b = ctx.box_fit(elev, o["grid"], o["num_cluster"])
    ob = oracle.box_fit(p, elev, o["grid"], o["num_cluster"])
    assert b["n_undefined"] == ob["n_undefined"]
    assert np.array_equal(b["box_cluster"], ob["box_cluster"])
    assert b["boxes"].shape == ob["boxes"].shape and np.array_equal(b["boxes"].view(np.uint32), ob["boxes"].view(np.uint32))
    return r, b


@pytest.mark.parametrize("n,stream,frame", [(120000, 0, 0), (120000, 1, 4), (120000, 2, 9), (200000, 4, 1), (30000, 5, 0), (5000, 6, 2)])
def test_cluster_box_parity(ctx, oracle, synth, n, stream, frame):
    p = oracle.params(0)
    cloud = synth.make_cloud(n, stream, frame)
    elev = oracle.ground_remove(p, cloud)["elevated"]
    _stage_parity(ctx, oracle, p, elev)


def test_cluster_box_small_and_empty(ctx, oracle, synth):
    p = oracle.params(0)
    _stage_parity(ctx, oracle, p, np.zeros((0, 4), np.float32))
    _stage_parity(ctx, oracle, p, np.array([[1, 1, 0, 0]], np.float32))
    two = np.array([[1, 1, 0, 0], [1.01, 1.01, 0.5, 0]], np.float32)   # one cell with 2 points -> 3x3 cluster, 2 pts < 30
    _stage_parity(ctx, oracle, p, two)
    edge = synth.edge_case_points()
    _stage_parity(ctx, oracle, p, np.concatenate([edge, edge]))


def interleaved_clusters_cloud(n_clusters=12, reps=40):
    """consecutive points hop between many small clusters: more distinct clusters per 64-point tile than the label
    kernel's per-tile summary holds, so the consumers' fallback path runs"""
    rng = np.random.default_rng(4)
    centres = [(-20 + 3.5 * k, -18 + 3.1 * k) for k in range(n_clusters)]
    pts = []
    for r in range(reps):
        for k, (cx, cy) in enumerate(centres):
            pts.append((cx + rng.uniform(-0.6, 0.6), cy + rng.uniform(-0.25, 0.25), rng.uniform(-1.0, 0.4), 0.0))
    return np.array(pts, np.float32)


def shuffled_many_clusters_cloud(n_clusters=60, per=340, seed=9):
    """random point order over many clusters: far more (tile, cluster) groups than the index kernel keeps in LDS"""
    rng = np.random.default_rng(seed)
    pts = []
    for k in range(n_clusters):
        cx, cy = -22 + 5.5 * (k % 8), -22 + 5.5 * (k // 8)
        p = np.zeros((per, 4), np.float32)
        p[:, 0] = cx + rng.uniform(-0.9, 0.9, per); p[:, 1] = cy + rng.uniform(-0.5, 0.5, per); p[:, 2] = rng.uniform(-1.2, 0.3, per)
        pts.append(p)
    pts = np.concatenate(pts)
    rng.shuffle(pts)
    return pts


def test_shuffled_many_clusters(ctx, oracle):
    p = oracle.params(0)
    cloud = shuffled_many_clusters_cloud()
    r, b = _stage_parity(ctx, oracle, p, cloud)
    assert r["num_cluster"] >= 40 and len(b["boxes"]) >= 10


def test_unordered_cloud_beyond_one_tile_window(mot, hip_lib, oracle):
    """138 000 elevated points in no order on the MI355X: the index kernel's many-groups path, two windows of its per-cluster tile table (round 6)"""
    p = oracle.params(0)
    cloud = shuffled_many_clusters_cloud(60, 2300, seed=4)
    with mot.Context(max_points=294912) as c:
        r, b = _stage_parity(c, oracle, p, cloud)
    assert r["num_cluster"] == 60 and len(b["boxes"]) > 0


def test_many_clusters_per_tile(ctx, oracle):
    p = oracle.params(0)
    _stage_parity(ctx, oracle, p, interleaved_clusters_cloud())
    _stage_parity(ctx, oracle, p, interleaved_clusters_cloud(20, 25))


def test_more_than_64_label_chunks(ctx, oracle, synth):
    """> 131072 elevated points = more 2048-point chunks than the index kernel's table prefix holds in LDS: general path"""
    p = oracle.params(0)
    elev = np.concatenate([oracle.ground_remove(p, synth.make_cloud(120000, s, f))["elevated"] for s, f in ((0, 0), (0, 1), (1, 0), (1, 2), (2, 1), (2, 3), (3, 0))])
    assert 131072 < len(elev) <= 262144
    _stage_parity(ctx, oracle, p, elev)


@pytest.mark.parametrize("n,stream,frame", [(120000, 0, 0), (200000, 4, 1), (9000, 6, 2)])
def test_side_products(ctx, oracle, synth, n, stream, frame):
    """cluster-node side products (clustered cloud, obstacle list, cost map): bit-exact against the restatement"""
    p = oracle.params(0)
    elev = np.concatenate([oracle.ground_remove(p, synth.make_cloud(n, stream, frame))["elevated"], synth.edge_case_points()])
    r = ctx.cluster(elev)
    a = ctx.cluster_products(0); o = oracle.cluster_products(p, elev, r["grid"])
    assert len(o["clustered"]) > 100 and len(o["obstacles"]) > 10
    for k in ("clustered", "obstacles", "cost_map"):
        assert a[k].shape == o[k].shape and np.array_equal(a[k], o[k]), k


def test_side_products_golden_and_empty(ctx, oracle):
    fx = G.load("side_ot_9k.npz")
    ctx.cluster(fx["elevated"])
    a = ctx.cluster_products(0)
    for k in ("clustered", "obstacles", "cost_map"):
        assert np.array_equal(a[k], fx[k]), k
    ctx.cluster(np.zeros((0, 4), np.float32))
    a = ctx.cluster_products(0)
    assert len(a["clustered"]) == 0 and len(a["obstacles"]) == 0 and not a["cost_map"].any()
    # the host-input form (the reference functions' own argument lists), with a non-default cost map
    sp = oracle.side_params(); sp.cost_width = 64; sp.cost_height = 40; sp.cost_resolution = 0.5; sp.cost_offset_x = 3.0
    import ctypes as C
    msp = type(ctx).cluster_products_host.__globals__["MotSideParams"]()
    C.memmove(C.byref(msp), C.byref(sp), C.sizeof(sp))
    a = ctx.cluster_products_host(fx["elevated"], fx["grid"].astype(np.int32), msp)
    o = oracle.cluster_products(oracle.params(0), fx["elevated"], fx["grid"].astype(np.int32), sp)
    for k in ("clustered", "obstacles", "cost_map"):
        assert a[k].shape == o[k].shape and np.array_equal(a[k], o[k]), k


def test_ccl_adversarial_patterns(mot, hip_lib, oracle):
    rng = np.random.default_rng(0)
    for preset in (0, 1):
        p = oracle.params(preset)
        with mot.Context(mot.params(preset), max_points=150000) as c:
            for name, cells in patterns.occupancy_cases(p.num_grid, rng):
                pts = patterns.case_points(cells, p, rng)
                r = c.cluster(pts)
                o = oracle.cluster(p, pts)
                assert r["num_cluster"] == o["num_cluster"], (preset, name)
                assert np.array_equal(r["grid"], o["grid"]), (preset, name)
                assert np.array_equal(r["point_label"], o["point_label"]), (preset, name)


def test_box_kitti_preset(mot, hip_lib, oracle, synth):
    p = oracle.params(1)
    with mot.Context(mot.params(1), max_points=131072) as c:
        for stream in (0, 3):
            elev = oracle.ground_remove(p, synth.make_cloud(120000, stream, 0))["elevated"]
            _stage_parity(c, oracle, p, elev)


@pytest.mark.parametrize("name", G.FRAMES)
def test_golden_frames(ctx, name):
    """fixtures produced by the reference's own sources (tests/golden/make_golden.py)"""
    fx = G.load(name)
    g = ctx.ground_remove(fx["cloud"])
    cl = ctx.cluster(g["elevated"])
    bx = ctx.box_fit(g["elevated"], cl["grid"], cl["num_cluster"])
    G.check_frame(fx, g, cl, bx["boxes"])


@pytest.mark.parametrize("name", G.FRAMES_OT0)
def test_golden_frames_ot0(mot, hip_lib, name):
    """the KITTI-tuned preset against the fixture produced by object_tracking0's own sources"""
    fx = G.load(name)
    with mot.Context(mot.params(1), max_points=65536) as c:
        g = c.ground_remove(fx["cloud"])
        cl = c.cluster(g["elevated"])
        bx = c.box_fit(g["elevated"], cl["grid"], cl["num_cluster"])
        G.check_frame(fx, g, cl, bx["boxes"])


def test_fused_frames_dev(ctx, oracle, synth):
    """ground -> cluster -> box for 8 frames in one launch sequence, everything resident in HBM"""
    import hiprt
    p = oracle.params(0)
    sizes = [120000, 1, 99999, 0, 200000, 2048, 77777, 131072]
    stride = 262144
    host = np.zeros((8, stride, 4), np.float32)
    clouds = []
    for b, n in enumerate(sizes):
        c = synth.make_cloud(max(n, 1), 20 + b, b)[:n]
        host[b, :n] = c
        clouds.append(c)
    dev = hiprt.DeviceBuffer(host)
    for rep in range(2):
        ctx.frames_dev(dev.ptr, stride * 4, sizes)
        for b, n in enumerate(sizes):
            g = oracle.ground_remove(p, clouds[b])
            r = ctx.get_ground(b, n_hint=n)
            assert np.array_equal(r["elevated"], g["elevated"]) and np.array_equal(r["ground"], g["ground"])
            assert np.array_equal(r["mask"][:n], g["mask"])
            o = oracle.cluster(p, g["elevated"])
            cl = ctx.get_clusters(b, n_elevated=len(g["elevated"]))
            assert cl["num_cluster"] == o["num_cluster"] and np.array_equal(cl["grid"], o["grid"])
            assert np.array_equal(cl["point_label"], o["point_label"])
            ob = oracle.box_fit(p, g["elevated"], o["grid"], o["num_cluster"])
            bx = ctx.get_boxes(b)
            assert np.array_equal(bx["boxes"], ob["boxes"]) and np.array_equal(bx["box_cluster"], ob["box_cluster"])
            if rep == 1 and b in (0, 3, 4):   # the cluster node's side products of a slot of the fused path
                sd = ctx.cluster_products(b); osd = oracle.cluster_products(p, g["elevated"], o["grid"])
                for k in ("clustered", "obstacles", "cost_map"):
                    assert sd[k].shape == osd[k].shape and np.array_equal(sd[k], osd[k]), (b, k)
    dev.free()


def test_fused_irregular_frames(mot, hip_lib, oracle, synth):
    """several hundred tiny clusters (speckled occupancy lists), and a frame with more than 65536 elevated points (many compaction
    and label chunks, long running positions), next to normal frames in the same batch"""
    import hiprt
    import irregular_clouds as ic
    p = oracle.params(0)
    many = ic.many_clusters_cloud(); crowded = ic.crowded_cloud(oracle, synth, 120000, 4); normal = synth.make_cloud(120000, 4, 0)
    assert oracle.cluster(p, oracle.ground_remove(p, many)["elevated"])["num_cluster"] > 255
    assert len(oracle.ground_remove(p, crowded)["elevated"]) > 65536
    clouds = [normal, many, crowded, normal[::-1].copy()]
    stride = ((max(len(x) for x in clouds) + 2047) // 2048) * 2048
    bufs = []
    def upload(host):
        d = hiprt.DeviceBuffer(host); bufs.append(d); return d.ptr, d
    with mot.Context(max_points=stride, max_batch=4) as c:
        seen = ic.check_fused_against_oracle(c, oracle, clouds, stride, upload)
    for d in bufs:
        d.free()
    assert seen["clusters"] > 255 and seen["boxes"] > 0


def test_full_size_properties(ctx, synth):
    """200k-point frame: labels only on occupied cells, ids contiguous, boxes in cluster order, idempotent"""
    cloud = synth.make_cloud(200000, 11, 0)
    g = ctx.ground_remove(cloud)
    cl = ctx.cluster(g["elevated"])
    ids = np.unique(cl["grid"])
    assert ids[0] == 0 and np.array_equal(ids[1:], np.arange(1, cl["num_cluster"] + 1))
    flat = cl["grid"].ravel()
    first = [np.argmax(flat == k) for k in range(1, cl["num_cluster"] + 1)]
    assert all(a < b for a, b in zip(first, first[1:]))
    bx = ctx.box_fit(g["elevated"], cl["grid"], cl["num_cluster"])
    assert np.all(np.diff(bx["box_cluster"]) > 0)
    cl2 = ctx.cluster(g["elevated"])
    assert np.array_equal(cl2["grid"], cl["grid"])
    bx2 = ctx.box_fit(g["elevated"], cl["grid"], cl["num_cluster"])
    assert np.array_equal(bx2["boxes"], bx["boxes"])


def test_box_fit_resident(ctx, oracle, synth):
    """the box stage on what mot_cluster left resident (the `cluster` node shell: one upload for both stages)"""
    p = oracle.params(0)
    e = oracle.ground_remove(p, synth.make_cloud(120000, 3, 1))["elevated"]
    cl = ctx.cluster(e)
    a = ctx.box_fit_resident()
    b = ctx.box_fit(e, cl["grid"], cl["num_cluster"])
    o = oracle.box_fit(p, e, oracle.cluster(p, e)["grid"], cl["num_cluster"])
    assert len(o["boxes"]) > 0 and np.array_equal(a["boxes"], o["boxes"]) and np.array_equal(b["boxes"], o["boxes"])
    assert np.array_equal(a["box_cluster"], b["box_cluster"])

@pytest.mark.parametrize("n,stream,frame", [(120000, 3, 1), (200000, 4, 1), (30000, 5, 0)])
def test_box_markers(ctx, oracle, synth, n, stream, frame):
    """the rviz cubes (mark_cluster, box_fitting.cpp:161-209) folded on the device: bit-equal to the float32 sums in input order, and to
    the MarkerArray the reference's own boxFitting fills where that build is present; then every slot of a fused batch"""
    import oracle_lib as O
    p = oracle.params(0)
    e = oracle.ground_remove(p, synth.make_cloud(n, stream, frame))["elevated"]
    cl = ctx.cluster(e)
    b = ctx.box_fit_resident()
    m = ctx.box_markers(0)
    assert len(m) == len(b["boxes"]) > 0
    assert np.array_equal(m.view(np.uint32), O.box_markers_numpy(e, cl["point_label"], b["box_cluster"]).view(np.uint32))
    if O.ref() is not None:
        mine = m.astype(np.float64); mine[:, 3:][mine[:, 3:] == 0] = 0.1
        assert np.array_equal(mine, O.ref_box_markers(e, cl["grid"], cl["num_cluster"]))
    with pytest.raises(Exception):
        ctx.box_markers(0, max_boxes=len(m) - 1)
    if n == 120000:
        import hiprt
        clouds = [synth.make_cloud(120000, s, f) for s, f in ((0, 0), (1, 4), (2, 9))]
        stride = 131072
        host = np.zeros((3, stride, 4), np.float32)
        for s, c in enumerate(clouds):
            host[s, : len(c)] = c
        dev = hiprt.DeviceBuffer(host)
        ctx.frames_dev(dev.ptr, stride * 4, [len(c) for c in clouds])
        for s, c in enumerate(clouds):
            es = oracle.ground_remove(p, c)["elevated"]
            k = ctx.get_clusters(s, len(es)); bb = ctx.get_boxes(s)
            assert np.array_equal(ctx.box_markers(s).view(np.uint32), O.box_markers_numpy(es, k["point_label"], bb["box_cluster"]).view(np.uint32))


def test_box_markers_golden(ctx):
    """the cube markers of the golden frame tests/golden/markers_ot_24k.npz (the reference's own boxFitting, make_golden.py) on the MI355X"""
    fx = G.load("markers_ot_24k.npz")
    cl = ctx.cluster(fx["elevated"])
    assert cl["num_cluster"] == int(fx["num_cluster"]) and np.array_equal(cl["grid"], fx["grid"].astype(np.int32))
    b = ctx.box_fit_resident()
    assert np.array_equal(b["boxes"].view(np.uint32), fx["boxes"].view(np.uint32))
    m = ctx.box_markers(0).astype(np.float64); m[:, 3:][m[:, 3:] == 0] = 0.1
    assert np.array_equal(m, fx["markers"])


ZERO_HEIGHT_CASES = ([np.nan, np.nan, np.nan, -0.0, np.nan, np.nan], [np.nan, 0.0, np.nan, -0.0, np.nan, np.nan], [-1.0, -0.0, 0.0, -0.5, np.nan, np.nan],
                     [-1.0, -2.0, -0.0, -0.0, 0.0, np.nan], [0.5, -0.0, 0.0, np.nan, np.nan, np.nan], [-0.0] * 6, [0.0] + [-0.0] * 5)


def _zero_height_cloud(zs):
    pts = [(0.0, 0.0, z) for z in zs[:4]] + [(0.0, 0.5, zs[4]), (0.5, 0.0, zs[5])]
    a = np.zeros((len(pts), 4), np.float32); a[:, :3] = np.array(pts, np.float32)
    return np.repeat(a, 5, axis=0)


def test_box_height_sign_of_zero(ctx, oracle):
    """`if (pZ > maxZ) maxZ = pZ` keeps the first of equal maxima; -0 and +0 are equal: the box's top face carries the sign of
    the FIRST zero of the cluster"""
    p = oracle.params(0)
    for zs in ZERO_HEIGHT_CASES:
        e = _zero_height_cloud(zs)
        o = oracle.cluster(p, e)
        b = ctx.box_fit(e, o["grid"], o["num_cluster"]); ob = oracle.box_fit(p, e, o["grid"], o["num_cluster"])
        assert len(ob["boxes"]) == 1 and np.array_equal(b["boxes"].view(np.uint32), ob["boxes"].view(np.uint32)), zs


@pytest.mark.parametrize("preset", [0, 1])
def test_large_irregular_clouds(mot, hip_lib, oracle, preset):
    """tests/test_emu_large_random.py's generator on the real kernels: scatter, blobs, walls, exact duplicates, boundary-snapped
    points, shuffled order — the multi-chunk machinery on data the HDL-64E scenes do not produce"""
    import test_emu_large_random as LR
    p = oracle.params(preset)
    with mot.Context(mot.params(preset), max_points=131072) as c:
        for seed in range(100 * preset, 100 * preset + 3):
            cloud = LR.big_cloud(seed)
            g = c.ground_remove(cloud); og = oracle.ground_remove(p, cloud)
            assert np.array_equal(g["mask"], og["mask"]) and np.array_equal(g["elevated"], og["elevated"]) and np.array_equal(g["ground"], og["ground"]), seed
            cl = c.cluster(og["elevated"]); ocl = oracle.cluster(p, og["elevated"])
            assert cl["num_cluster"] == ocl["num_cluster"] and np.array_equal(cl["grid"], ocl["grid"]) and np.array_equal(cl["point_label"], ocl["point_label"]), seed
            if ocl["num_cluster"] > 4096:
                continue
            bx = c.box_fit_resident(); obx = oracle.box_fit(p, og["elevated"], ocl["grid"], ocl["num_cluster"])
            assert bx["n_undefined"] == obx["n_undefined"] and np.array_equal(bx["box_cluster"], obx["box_cluster"]), seed
            assert np.array_equal(bx["boxes"].view(np.uint32), obx["boxes"].view(np.uint32)), seed
            sd = c.cluster_products(0); osd = oracle.cluster_products(p, og["elevated"], ocl["grid"])
            for k in ("clustered", "obstacles", "cost_map"):
                assert sd[k].shape == osd[k].shape and np.array_equal(sd[k], osd[k]), (seed, k)


def test_wide_clusters_take_the_large_hull_kernel(mot, hip_lib, oracle):
    """see tests/test_emu_cluster_box.py: the per-frame flag for cluster_rect_large_kernel, on the MI355X (stage-wise calls and a fused batch
    that mixes frames with and without wide clusters)"""
    import hiprt
    import wide_clusters as W
    kw = dict(t_len_max=100.0, t_area_max=200.0, t_width_max=10.0, t_ratio_max=500.0, t_pt_per_m3=0.1)
    p = oracle.params(0, **kw)
    with mot.Context(mot.params(0, **kw), max_points=32768, max_batch=4) as c:
        wide = 0
        for seed, walls in ((0, 2), (1, 0), (2, 1), (3, 0), (4, 2)):
            bx, got = W.check(c, oracle, p, W.wide_wall_cloud(seed, walls))
            wide += sum(1 for d in bx["debug"] if d["branch"] == 1 and d["num_points"] > 5000 and d["accepted"])   # a box that only the large-hull kernel can have produced
        assert wide >= 4


def test_l_shape_cluster_with_more_groups_than_the_staging_holds(mot, hip_lib, oracle):
    """the L-shape branch finds "the r-th point of the cluster" by a search over the cluster's (tile, cluster) groups; up to 512 of them
    are staged in LDS, a cluster with more searches global memory: a 40 000-point L in random order (~640 groups), both RNG mappings"""
    import wide_clusters as W
    for mapping in (1, 0):   # MOT_RNG_LIBSTDCXX11 (default), MOT_RNG_LIBSTDCXX10
        kw = dict(rng_mapping=mapping)
        p = oracle.params(0, **kw)
        with mot.Context(mot.params(0, **kw), max_points=65536) as c:
            for seed in (1, 2):
                bx, got = W.check(c, oracle, p, W.big_l_cloud(seed))
                assert any(d["branch"] == 0 and d["num_points"] >= 33000 and d["accepted"] for d in bx["debug"]), [(d["num_points"], d["branch"], d["accepted"]) for d in bx["debug"]]


def test_node_frame_calls(ctx, oracle, synth):
    """mot_cluster_node_frame / mot_ground_node_frame on the MI355X = the call-by-call sequences they bundle (tests/node_frame_case.py)"""
    import node_frame_case
    node_frame_case.check(ctx, oracle, synth, sizes=((120000, 3, 1), (30000, 5, 0), (200000, 4, 1)))
