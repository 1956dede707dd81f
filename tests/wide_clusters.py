"""clouds with clusters whose min-area-rectangle branch has MORE candidate hull points than cluster_rect_kernel takes (510): they go to
cluster_rect_large_kernel, which runs only in frames the gather kernel flags. Shared by the emulator and GPU box tests."""
import numpy as np


def wide_wall_cloud(seed=0, walls=2, n_small=3):
    """elevated points only: `walls` long thin diagonal-free walls (hundreds of pixel columns, two extreme pixels each) and a few
    compact blobs; slopes arranged so that the walls take the rectangle branch (short slope distance) — plus, in the same cloud, a
    frame WITHOUT wide clusters is obtained with walls=0"""
    rng = np.random.default_rng(seed)
    pts = []
    for w in range(walls):
        n = 9000
        x = rng.uniform(-22.0, 22.0, n)                       # 44 m along x: ~790 picture columns
        y = (9.0 + 4.0 * w) + rng.uniform(-0.45, 0.45, n)      # thick enough for distinct lowest / highest pixel rows per column
        z = rng.uniform(-1.2, 0.4, n)                           # height 2.4 m: inside the rule filter
        pts.append(np.stack([x, y, z, np.ones(n)], 1))
    for k in range(n_small):
        n = 600
        c = rng.uniform(-15, 15, 2) * [1, 0.3] + [0, -8 - 3 * k]
        pts.append(np.concatenate([c + rng.normal(0, 0.5, (n, 2)), rng.uniform(-1.2, 0.6, (n, 1)), np.ones((n, 1))], 1))
    a = np.concatenate(pts).astype(np.float32)
    return a[rng.permutation(len(a))] if seed % 2 else a


def check(ctx, oracle, p, cloud):
    cl = oracle.cluster(p, cloud); bx = oracle.box_fit(p, cloud, cl["grid"], cl["num_cluster"], debug=True)
    a = ctx.cluster(cloud); b = ctx.box_fit(cloud, a["grid"], a["num_cluster"])
    assert a["num_cluster"] == cl["num_cluster"] and np.array_equal(a["grid"], cl["grid"])
    assert np.array_equal(b["boxes"].view(np.uint32), bx["boxes"].view(np.uint32)) and np.array_equal(b["box_cluster"], bx["box_cluster"]) and b["n_undefined"] == bx["n_undefined"]
    # the rviz cubes of the same boxes (mot_box_markers): clusters of thousands of points fill the kernel's 2048-point LDS window several
    # times over, in random order they are hundreds of sparse tiles
    import oracle_lib as O
    m = ctx.box_markers(0)
    assert np.array_equal(m.view(np.uint32), O.box_markers_numpy(cloud, cl["point_label"], bx["box_cluster"]).view(np.uint32))
    return bx, b


def big_l_cloud(seed=1, n_per_wall=20000, n_small=2):
    """one L-shaped cluster of 2 x n_per_wall points (the two visible sides of a car, very densely sampled) in RANDOM order, so that almost every
    64-point tile holds points of it: more (tile, cluster) groups than the gather kernel stages in LDS for the L-shape branch's
    "r-th point of the cluster" search (512) — plus a few compact blobs"""
    rng = np.random.default_rng(seed)
    n = n_per_wall
    a = np.stack([rng.uniform(5.0, 9.0, n), 10.0 + rng.uniform(-0.1, 0.1, n), rng.uniform(-1.2, 0.0, n), np.ones(n)], 1)      # a car's long side ...
    b = np.stack([9.0 + rng.uniform(-0.1, 0.1, n), rng.uniform(10.0, 11.8, n), rng.uniform(-1.2, 0.0, n), np.ones(n)], 1)     # ... and its short side
    pts = [a, b]
    for k in range(n_small):
        m = 700
        c = np.array([-10.0 - 6 * k, -12.0])
        pts.append(np.concatenate([c + rng.normal(0, 0.5, (m, 2)), rng.uniform(-1.2, 0.6, (m, 1)), np.ones((m, 1))], 1))
    out = np.concatenate(pts).astype(np.float32)
    return out[rng.permutation(len(out))]
