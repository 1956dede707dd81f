"""Large irregular clouds (tens of thousands of points: uniform scatter, dense blobs, walls, exact duplicates, points on the
ring / ROI / cell boundaries, shuffled order) through the whole stateless chain under the emulator against the restatement —
the multi-chunk machinery (look-back compaction across chunks, (tile, cluster) groups, cluster tables, the irregular-chunk
fallback of the index kernel) on data the synthetic HDL-64E scenes do not produce. MOT_PROP_SCALE multiplies the count."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
SCALE = int(os.environ.get("MOT_PROP_SCALE", "1"))


def big_cloud(seed):
    rng = np.random.default_rng(seed)
    parts = []
    n_uni = int(rng.integers(2000, 30000))
    r = rng.uniform(0, 60, n_uni) ** rng.choice([1.0, 0.5]); th = rng.uniform(-np.pi, np.pi, n_uni)
    parts.append(np.stack([r * np.cos(th), r * np.sin(th), rng.uniform(-2.5, 1.5, n_uni)], 1))
    for _ in range(int(rng.integers(0, 40))):        # blobs: cars / poles / bushes
        c = rng.uniform(-28, 28, 2); s = rng.uniform(0.05, 2.5, 2); k = int(rng.integers(5, 1500))
        parts.append(np.stack([rng.normal(c[0], s[0], k), rng.normal(c[1], s[1], k), rng.uniform(-1.9, rng.uniform(-1.5, 1.0), k)], 1))
    for _ in range(int(rng.integers(0, 6))):         # walls
        a, b = rng.uniform(-30, 30, 2), rng.uniform(-30, 30, 2); k = int(rng.integers(100, 4000)); t = rng.random(k)[:, None]
        parts.append(np.concatenate([a + t * (b - a) + rng.normal(0, 0.03, (k, 2)), rng.uniform(-1.9, 0.8, (k, 1))], 1))
    pts = np.concatenate(parts).astype(np.float32)
    if rng.random() < 0.5:                            # snap some points onto cell / ROI / ring boundaries
        k = len(pts) // 20; idx = rng.integers(0, len(pts), k)
        pts[idx, 0] = (np.round(pts[idx, 0] / 0.2) * 0.2).astype(np.float32)
        pts[idx[: k // 4], 1] = rng.choice(np.array([25.0, -25.0, 24.999998, 3.4, -3.4, 0.0], np.float32), k // 4)
    if rng.random() < 0.5:
        pts = np.concatenate([pts, pts[rng.integers(0, len(pts), len(pts) // 10)]])   # exact duplicates
    if rng.random() < 0.7:
        rng.shuffle(pts)
    out = np.zeros((len(pts), 4), np.float32); out[:, :3] = pts
    return out


@pytest.mark.parametrize("preset", [0, 1])
def test_emu_large_irregular_clouds(mot, oracle, preset):
    import build_emu
    lib = build_emu.build()
    p = oracle.params(preset)
    with mot.Context(mot.params(preset, lib=mot.load_library(lib)), lib_path=lib, max_points=131072) as c:
        for seed in range(100 * preset, 100 * preset + 3 * SCALE):
            cloud = big_cloud(seed)
            g = c.ground_remove(cloud); og = oracle.ground_remove(p, cloud)
            assert np.array_equal(g["mask"], og["mask"]) and np.array_equal(g["elevated"], og["elevated"]) and np.array_equal(g["ground"], og["ground"]), seed
            cl = c.cluster(og["elevated"]); ocl = oracle.cluster(p, og["elevated"])
            assert cl["num_cluster"] == ocl["num_cluster"] and np.array_equal(cl["grid"], ocl["grid"]) and np.array_equal(cl["point_label"], ocl["point_label"]), seed
            if ocl["num_cluster"] > 4096:
                continue   # beyond the library's cluster capacity: MOT_E_CAPACITY, covered elsewhere
            bx = c.box_fit_resident(); obx = oracle.box_fit(p, og["elevated"], ocl["grid"], ocl["num_cluster"])
            assert bx["n_undefined"] == obx["n_undefined"] and np.array_equal(bx["box_cluster"], obx["box_cluster"]), seed
            assert np.array_equal(bx["boxes"].view(np.uint32), obx["boxes"].view(np.uint32)), seed
            sd = c.cluster_products(0); osd = oracle.cluster_products(p, og["elevated"], ocl["grid"])
            for k in ("clustered", "obstacles", "cost_map"):
                assert sd[k].shape == osd[k].shape and np.array_equal(sd[k], osd[k]), (seed, k)


def test_emu_ragged_batch_of_irregular_clouds(mot, oracle):
    """mot_frames_dev on a batch of frames of very different sizes (one of them empty, one tiny) with the tracker on: every
    slot must equal the restatement run on that frame alone"""
    import build_emu
    lib = build_emu.build()
    p = oracle.params(0)
    B, stride = 4, 65536
    with mot.Context(lib_path=lib, max_points=stride, max_batch=B, max_tracks_total=512) as c:
        Ts = [oracle.Tracker(p) for _ in range(B)]
        for f in range(2 * SCALE):
            clouds = [big_cloud(700 + 10 * f + b)[: [stride, 0, 37, 9000][b]] for b in range(B)]
            host = np.zeros((B, stride, 4), np.float32)
            for b in range(B):
                host[b, : len(clouds[b])] = clouds[b]
            ts = [2.0e8 + f * 1e5] * B
            c.frames_dev(host.ctypes.data, stride * 4, [len(x) for x in clouds], run_tracker=True, timestamps=ts, ego_v=[1.0] * B, ego_yaw=[0.0] * B)
            for b in range(B):
                og = oracle.ground_remove(p, clouds[b]); ocl = oracle.cluster(p, og["elevated"])
                obx = oracle.box_fit(p, og["elevated"], ocl["grid"], ocl["num_cluster"])["boxes"]
                g = c.get_ground(b)
                assert np.array_equal(g["elevated"], og["elevated"]) and np.array_equal(g["ground"], og["ground"]), (f, b)
                assert np.array_equal(c.get_boxes(b)["boxes"], obx), (f, b)
                ego = Ts[b].ego_update(ts[b], 1.0, 0.0)
                co, si = np.cos(-ego[2]), np.sin(-ego[2])
                gb = obx.astype(np.float64).copy()
                dx, dy = gb[..., 0] - ego[0], gb[..., 1] - ego[1]
                gb[..., 0] = co * dx - si * dy; gb[..., 1] = si * dx + co * dy
                o = Ts[b].step(gb.astype(np.float32), ts[b]); a = c.get_tracks(b)
                assert a["n"] == o["n"] and np.array_equal(a["track_manage"], o["track_manage"]), (f, b)


def test_emu_fused_irregular_frames(mot, oracle, synth):
    """several hundred tiny clusters, and more than 65536 elevated points in one frame, next to a normal frame in the same batch"""
    import build_emu
    import irregular_clouds as ic
    lib = build_emu.build()
    p = oracle.params(0)
    many = ic.many_clusters_cloud(); crowded = ic.crowded_cloud(oracle, synth, 60000, 9); normal = synth.make_cloud(9000, 4, 0)
    assert oracle.cluster(p, oracle.ground_remove(p, many)["elevated"])["num_cluster"] > 255
    assert len(oracle.ground_remove(p, crowded)["elevated"]) > 65536
    clouds = [many, crowded, normal]
    stride = max(len(x) for x in clouds) + 64
    with mot.Context(lib_path=lib, max_points=stride, max_batch=3) as c:
        seen = ic.check_fused_against_oracle(c, oracle, clouds, stride, lambda host: (host.ctypes.data, host))
    assert seen["clusters"] > 255
