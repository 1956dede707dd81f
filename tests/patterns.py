"""adversarial occupancy patterns for the connected-component stage (cells -> points that fall in them)"""
import numpy as np


def cells_to_points(cells, G, roi, reps, rng):
    cs = np.array(cells, np.float64).reshape(-1, 2)
    xy = (cs + 0.5) * (roi / G) - roi / 2
    pts = np.repeat(xy, reps, axis=0)
    out = np.zeros((len(pts), 4), np.float32)
    out[:, :2] = pts
    out[:, 2] = rng.uniform(-1, 1, len(pts))
    return out


def occupancy_cases(G, rng, dense=True):
    cases = []
    for dens in (0.002, 0.02, 0.1, 0.3, 0.6):
        m = rng.random((G, G)) < dens
        cases.append(("rand%.3f" % dens, [(x, y) for x in range(G) for y in range(G) if m[x, y]]))
    cases.append(("checker5", [(x, y) for x in range(0, G, 5) for y in range((x // 5) % 2 * 2, G, 5)]))
    cases.append(("hstripes", [(x, y) for x in range(0, G, 4) for y in range(G)]))
    cases.append(("vstripes", [(x, y) for x in range(G) for y in range(0, G, 4)]))
    cases.append(("diag", [(i, i) for i in range(G)] + [(i, G - 1 - i) for i in range(G)]))
    cases.append(("corners", [(0, 0), (0, G - 1), (G - 1, 0), (G - 1, G - 1)]))
    cases.append(("edges", [(0, y) for y in range(G)] + [(G - 1, y) for y in range(0, G, 7)] + [(x, 0) for x in range(0, G, 9)]))
    cases.append(("empty", []))
    sp = []; x = y = G // 2; dx, dy = 0, 1; seg = 1; k = 0
    while 0 <= x < G and 0 <= y < G and k < 4000:
        for _ in range(2):
            for _ in range(seg * 6):
                if 0 <= x < G and 0 <= y < G:
                    sp.append((x, y))
                x += dx; y += dy; k += 1
            dx, dy = dy, -dx
        seg += 1
    cases.append(("spiral", sp))
    cases.append(("comb", [(10, y) for y in range(5, G - 5)] + [(x, yy) for yy in range(5, G - 5, 6) for x in range(10, G - 10)]))
    if not dense:
        cases = [c for c in cases if c[0] in ("rand0.020", "rand0.300", "diag", "corners", "empty", "spiral", "comb", "edges")]
    return cases


def case_points(cells, p, rng):
    G, roi = p.num_grid, p.roi_m
    pts = cells_to_points(cells, G, roi, 2 if p.occ_min_count == 2 else 1, rng)
    single = cells_to_points([(int(a), int(b)) for a, b in rng.integers(0, G, size=(50, 2))], G, roi, 1, rng)
    pts = np.concatenate([pts, single])
    rng.shuffle(pts)
    return pts
