"""Seeded synthetic HDL-64E-like cloud generator.

No KITTI data ships with the reference or this image, so parity tests and bench.py run on this
generator: 64 beams (+2° … −24.8°), beam-major point order (as KITTI .bin files), ground plane at
z = −1.73 m with noise and a gentle tilt, 20–40 moving box obstacles (cars, pedestrians, walls),
1 % dropouts, intensity U(0,1).  Points are float32 (x, y, z, intensity), 16 B each — the layout of
``sensor_msgs/PointCloud2`` from kitti2bag / PCL ``PointXYZ`` that the reference's ``ground`` node
receives (OT/src/groundremove/main.cpp:100).
"""
from __future__ import annotations

import numpy as np

_KINDS = (
    (4.5, 1.8, 1.5),   # car
    (0.6, 0.6, 1.7),   # pedestrian
    (20.0, 0.3, 3.0),  # wall
)


def scene(stream: int, n_obstacles: int | None = None):
    """Static description of the obstacles of a stream: (centre xy, yaw, size lwh, velocity xy)."""
    rng = np.random.default_rng(7_000_003 + stream)
    k = int(rng.integers(20, 41)) if n_obstacles is None else n_obstacles
    kinds = rng.choice(3, size=k, p=(0.6, 0.25, 0.15))
    size = np.array([_KINDS[i] for i in kinds], dtype=np.float64)
    xy = rng.uniform(-40.0, 40.0, size=(k, 2))
    # keep the ego vehicle's own footprint free
    near = np.hypot(xy[:, 0], xy[:, 1]) < 4.0
    xy[near] += np.sign(xy[near] + 1e-9) * 5.0
    yaw = rng.uniform(-np.pi, np.pi, size=k)
    speed = np.where(kinds == 0, rng.uniform(0.0, 10.0, size=k), np.where(kinds == 1, rng.uniform(0.0, 1.5, size=k), 0.0))
    vel = np.stack([speed * np.cos(yaw), speed * np.sin(yaw)], axis=1)
    return xy, yaw, size, vel


def make_cloud(n_points: int = 120_000, stream: int = 0, frame: int = 0, dt: float = 0.1,
               n_obstacles: int | None = None) -> np.ndarray:
    """Return an (n_points, 4) float32 array. seed = 1000*stream + frame."""
    rng = np.random.default_rng(1000 * stream + frame)
    # ~12 % of the rays (the upward beams) see neither ground nor obstacle within 120 m and give no
    # return; oversample the azimuth so that ~n_points returns remain, then trim/pad to exactly n_points
    n_az = -(-int(n_points * 1.18) // 64)
    elev = np.deg2rad(np.linspace(2.0, -24.8, 64))
    az = np.linspace(-np.pi, np.pi, n_az, endpoint=False) + rng.uniform(0, 2 * np.pi / n_az)
    el = np.repeat(elev, n_az)                # beam-major
    a = np.tile(az, 64)
    d = np.stack([np.cos(el) * np.cos(a), np.cos(el) * np.sin(a), np.sin(el)], axis=1)

    # ground plane through (0,0,-1.73), tilted 1 deg about the y axis
    tilt = np.deg2rad(1.0)
    nrm = np.array([np.sin(tilt), 0.0, np.cos(tilt)])
    denom = d @ nrm
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(denom < -1e-6, (-1.73 * nrm[2]) / denom, np.inf)
    t = np.where(t * np.cos(el) < 121.0, t, np.inf)   # returns beyond ~121 m horizontal range are lost
    hit_ground = np.isfinite(t)

    xy, yaw, size, vel = scene(stream, n_obstacles)
    xy = xy + vel * (frame * dt)
    for k in range(len(xy)):
        c, s = np.cos(yaw[k]), np.sin(yaw[k])
        # ray in the box frame (origin = sensor at (0,0,0) world; box bottom at z=-1.73)
        ox = -(c * xy[k, 0] + s * xy[k, 1])
        oy = -(-s * xy[k, 0] + c * xy[k, 1])
        oz = 1.73 - size[k, 2] / 2.0
        dx = c * d[:, 0] + s * d[:, 1]
        dy = -s * d[:, 0] + c * d[:, 1]
        dz = d[:, 2]
        h = size[k] / 2.0
        with np.errstate(divide="ignore", invalid="ignore"):
            tx1, tx2 = (-h[0] - ox) / dx, (h[0] - ox) / dx
            ty1, ty2 = (-h[1] - oy) / dy, (h[1] - oy) / dy
            tz1, tz2 = (-h[2] - oz) / dz, (h[2] - oz) / dz
        tmin = np.maximum(np.maximum(np.minimum(tx1, tx2), np.minimum(ty1, ty2)), np.minimum(tz1, tz2))
        tmax = np.minimum(np.minimum(np.maximum(tx1, tx2), np.maximum(ty1, ty2)), np.maximum(tz1, tz2))
        ok = (tmax >= tmin) & (tmin > 0.5) & (tmin < t)
        t = np.where(ok, tmin, t)
        hit_ground &= ~ok

    got = np.isfinite(t)
    d, t, hit_ground = d[got], t[got], hit_ground[got]
    p = d * t[:, None]
    p[:, 2] += np.where(hit_ground, rng.normal(0.0, 0.02, size=len(t)), rng.normal(0.0, 0.005, size=len(t)))
    p[:, :2] += rng.normal(0.0, 0.003, size=(len(t), 2))
    keep = rng.random(len(t)) >= 0.01         # 1 % dropouts
    p = p[keep]
    inten = rng.random(len(p))
    out = np.concatenate([p, inten[:, None]], axis=1).astype(np.float32)
    if len(out) >= n_points:                   # thin uniformly, order preserved
        sel = np.sort(rng.choice(len(out), size=n_points, replace=False))
        out = out[sel]
    else:                                      # pad by repeating the head (keeps n exact)
        out = np.concatenate([out, out[: n_points - len(out)]], axis=0)
    return np.ascontiguousarray(out)


def edge_case_points(params_r_min: float = 3.4, params_r_max: float = 120.0, roi_m: float = 50.0) -> np.ndarray:
    """Hand-picked points on the boundaries the reference's comparisons sit on."""
    f = np.float32
    pts = [
        (params_r_min, 0, -1.7), (0, params_r_min, -1.7), (params_r_max, 0, -1.7), (0, -params_r_max, 0.5),
        (np.nextafter(f(params_r_min), f(10)), 0, -1.7), (np.nextafter(f(params_r_max), f(0)), 0, -1.7),
        (-10, 0.0, -1.7), (-10, -0.0, -1.7),          # atan2 = +pi / -pi  (chI = 80 dropped / 0)
        (10, 0, -1.7), (0, 10, -1.7), (0, -10, -1.7), (7, 7, -1.7), (-7, 7, -1.0), (7, -7, 0.3),
        (roi_m / 2, 1, 0.5), (-roi_m / 2, 1, 0.5), (1, roi_m / 2, 0.5), (1, -roi_m / 2, 0.5),
        (np.nextafter(f(roi_m / 2), f(0)), 1, 0.5), (0.0, 5.0, 0.2), (5.0, 0.0, 0.2),
        (3, 4, -0.4), (3, 4, -2.0), (6, 8, -0.4), (6, 8, -2.0), (0, 0, 0), (1e-20, 1e-20, 0),
        (200, 200, 0), (-3.4, 0, -1.9), (2.4041631, 2.4041631, -1.8),
    ]
    a = np.array([(x, y, z, 0.5) for x, y, z in pts], dtype=np.float32)
    return a
