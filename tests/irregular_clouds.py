"""irregular frames for the fused path (shared by the emulator and the GPU test): several hundred tiny clusters, and a frame
whose obstacle returns are repeated until more than 65536 points are elevated (many compaction chunks, many label chunks)"""
import numpy as np


def many_clusters_cloud(seed=5):
    """isolated blobs on every fifth cell of the cluster grid, three returns each: several hundred clusters, speckled occupancy"""
    import patterns
    rng = np.random.default_rng(seed)
    G, roi = 250, 50.0
    cells = [(x, y) for x in range(2, G - 2, 5) for y in range(2, G - 2, 5) if (x - G // 2) ** 2 + (y - G // 2) ** 2 > 18 ** 2]
    pts = patterns.cells_to_points(cells, G, roi, 3, rng)
    pts[:, 2] = rng.uniform(-0.9, 0.2, len(pts))
    return pts[rng.permutation(len(pts))]


def crowded_cloud(oracle, synth, n_base, copies, seed=3):
    """a scan whose obstacle returns are repeated `copies` times with millimetre jitter (same cells, same ground)"""
    base = synth.make_cloud(n_base, seed, 0)
    e = oracle.ground_remove(oracle.params(0), base)["elevated"]
    rng = np.random.default_rng(seed)
    extra = [e + np.concatenate([rng.normal(0, 0.002, (len(e), 2)), np.zeros((len(e), 2))], axis=1).astype(np.float32) for _ in range(copies)]
    return np.concatenate([base] + extra).astype(np.float32)


def check_fused_against_oracle(ctx, oracle, clouds, stride, upload):
    """one mot_frames_dev call over `clouds` (upload(host) -> device pointer or host pointer for the emulator), every slot against the oracle"""
    p = oracle.params(0)
    B = len(clouds)
    host = np.zeros((B, stride, 4), np.float32)
    for s, c in enumerate(clouds):
        host[s, : len(c)] = c
    ptr, keep = upload(host)
    ctx.frames_dev(ptr, stride * 4, [len(c) for c in clouds]); ctx.synchronize()
    seen = dict(clusters=0, boxes=0)
    for s, c in enumerate(clouds):
        g = oracle.ground_remove(p, c)
        r = ctx.get_ground(s, n_hint=len(c))
        assert np.array_equal(r["elevated"], g["elevated"]) and np.array_equal(r["ground"], g["ground"]) and np.array_equal(r["mask"][: len(c)], g["mask"])
        o = oracle.cluster(p, g["elevated"])
        cl = ctx.get_clusters(s, n_elevated=len(g["elevated"]))
        assert cl["num_cluster"] == o["num_cluster"] and np.array_equal(cl["grid"], o["grid"]) and np.array_equal(cl["point_label"], o["point_label"])
        ob = oracle.box_fit(p, g["elevated"], o["grid"], o["num_cluster"])
        bx = ctx.get_boxes(s)
        assert np.array_equal(bx["boxes"], ob["boxes"]) and np.array_equal(bx["box_cluster"], ob["box_cluster"])
        seen["clusters"] += o["num_cluster"]; seen["boxes"] += len(ob["boxes"])
    return seen
