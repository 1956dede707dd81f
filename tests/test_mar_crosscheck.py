""">= 10 000 proper random integer point sets (and ~3 000 degenerate ones) through tests/mar_check.py: the restated cv::minAreaRect against the
exhaustive integer oracle. The clusters of whole rendered streams go through the same check in tests/test_sequence_gpu.py and in
bench.py's parity_check (the renderer needs the GPU)."""
import os

import numpy as np

import mar_check as MC

SCALE = int(os.environ.get("MOT_PROP_SCALE", "1"))


def point_sets(rng, trials):
    for t in range(trials):
        n = int(rng.integers(1, 120)); mode = t % 10
        if mode == 0: p = rng.integers(-200, 200, (n, 2))
        elif mode == 1: p = rng.integers(0, 2500, (n, 2))                          # the picture's pixel range (picScale 30 x roiM 50 / 3)
        elif mode == 2: x = rng.integers(-300, 300, n); p = np.stack([x, x // 2 + rng.integers(-3, 3, n)], 1)     # thin, slanted (a wall)
        elif mode == 3: x = rng.integers(0, 400, n); p = np.stack([x, rng.integers(0, 12, n)], 1)                 # thin, axis-aligned
        elif mode == 4:                                                                 # an oriented box outline (a car seen from two sides)
            a = rng.uniform(0, np.pi); L, W = rng.uniform(20, 150), rng.uniform(10, 60); s = rng.uniform(0, 1, n)
            side = rng.integers(0, 2, n); q = np.where(side[:, None] == 0, np.stack([s * L, np.zeros(n)], 1), np.stack([np.zeros(n), s * W], 1))
            R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]); p = np.rint(q @ R.T + rng.uniform(500, 1500, 2)).astype(np.int64)
        elif mode == 5: p = rng.integers(-4, 4, (n, 2))                                  # tiny: many duplicates and ties
        elif mode == 6: k = int(rng.integers(3, 9)); p = rng.integers(-1000, 1000, (k, 2))   # few points
        elif mode == 7: x = rng.integers(-50, 50, n); p = np.stack([x, 3 * x + 7], 1)     # collinear
        elif mode == 8: p = np.stack([np.full(n, 7), rng.integers(-30, 30, n)], 1)        # one pixel column
        else:                                                                           # a disc: many hull edges, near-ties
            a = rng.uniform(0, 2 * np.pi, n); rad = rng.uniform(5, 300); p = np.rint(np.stack([np.cos(a), np.sin(a)], 1) * rad * np.sqrt(rng.uniform(0.5, 1, (n, 1)))).astype(np.int64)
        p = np.asarray(p, np.int32)
        if t % 4 == 1:
            p = np.concatenate([p, p[rng.integers(0, len(p), len(p) // 2 + 1)]])
        yield p


def test_restated_min_area_rect_against_the_exhaustive_oracle(oracle):
    rng = np.random.default_rng(20260926)
    worst, tied, proper = 0.0, 0, 0
    for t, p in enumerate(point_sets(rng, 14000 * SCALE)):
        rel, ties, clause = MC.check(oracle, p, where=t)
        assert clause == "rounding"          # none of these sets is thin enough for the cosine-resolution clause
        worst = max(worst, rel); tied += ties > 1; proper += ties > 0
    assert proper >= 10000 * SCALE and tied > 100    # real rectangles, and sets on which several hull edges attain the minimum exactly
    print("min-area rectangle cross-check: %d proper sets, %d with exact ties, worst area error %.2f x eps32 x diameter x (w + h)" % (proper, tied, worst))


def test_brute_oracle_on_known_shapes(oracle):
    """the exhaustive oracle itself on shapes whose answer is known in closed form"""
    sq = np.array([[0, 0], [10, 0], [10, 10], [0, 10], [5, 5]], np.int32)
    b = oracle.mar_brute(sq); assert b["min_area"] == 100.0 and len(b["hull"]) == 4 and b["ties"] == 4
    dia = np.array([[0, 5], [5, 0], [10, 5], [5, 10]], np.int32)          # a square standing on a corner: side 5 sqrt 2
    b = oracle.mar_brute(dia); assert abs(b["min_area"] - 50.0) < 1e-12 and b["ties"] == 4
    tri = np.array([[0, 0], [8, 0], [0, 6]], np.int32)                    # right triangle: legs give 48, the hypotenuse gives 10 x 4.8 = 48 too
    b = oracle.mar_brute(tri); assert abs(b["min_area"] - 48.0) < 1e-12 and b["ties"] == 3
    par = np.array([[0, 0], [10, 0], [13, 4], [3, 4]], np.int32)          # parallelogram: along the long edge 13 x 4 = 52; along the slanted one (|e| = 5): 11 x 8 = 88
    b = oracle.mar_brute(par); assert abs(b["min_area"] - 52.0) < 1e-12 and sorted(np.round(b["edge_area"], 6)) == [52.0, 52.0, 88.0, 88.0]
    seg = np.array([[1, 1], [4, 5], [1, 1]], np.int32)
    b = oracle.mar_brute(seg); assert len(b["hull"]) == 2 and b["min_area"] == 0.0
    big = np.array([[-30000, -30000], [30000, -29999], [30000, 30000], [-29999, 30000]], np.int32)   # 2^68-sized numerators
    b = oracle.mar_brute(big); assert np.isfinite(b["min_area"]) and abs(b["min_area"] - 3.6e9) < 2e5


def test_thin_walls_from_the_rendered_streams(oracle):
    """tests/golden/mar_thin_walls.npz: the 69 rectangle clusters of the rendered 154-frame streams (scenes 0, 1001 at 120 k points,
    scene 7 at 200 k; collected on the MI355X run of tests/test_sequence_gpu.py) on which the float32 caliper walk does NOT end on the
    exact minimum — 12 m walls 1.3 pixels thin. The double-precision walk does (asserted inside check), the float32 result stays within
    the cosine-resolution clause, and none of them can become a box (thin side < 2 pixels = 0.11 m against the rule filter's 0.2 m)."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mar_thin_walls.npz"))
    n = 0
    for k in d.files:
        rel, _, clause = MC.check(oracle, d[k], where=k)
        rr = oracle.min_area_rect(d[k])
        assert clause == "cosine resolution" and rel > MC.AREA_UNITS and min(rr[2], rr[3]) < 2.0
        n += 1
    assert n == 69
