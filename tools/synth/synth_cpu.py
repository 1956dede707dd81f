"""numpy mirror of tools/synth/synth.hip's ray caster (same scene generators, same geometry; its own noise stream, so NOT bit-equal to
the GPU renderer's frames): lets the tracker studies that need a rendered street / plaza sequence run on a box without a GPU
(tests/track_parity_study.py). BENCH / TEST DATA GENERATOR, not product code."""
from __future__ import annotations

import numpy as np

from synth_dev import ego_path, load_ego, plaza_scene, street_scene   # noqa: F401  (same directory on sys.path)

SENSOR_Z, MAX_RANGE = np.float32(1.73), np.float32(121.0)


def render_frame(objs: np.ndarray, path: np.ndarray, frame: int, n_points: int, dt: float = 0.1, seed: int = 2025, scene_id: int = 0,
                 oversample: float = 1.25, order: str = "beam") -> np.ndarray:
    """one frame of a scene: (n, 4) float32, n <= n_points, thinned uniformly in beam-major ray order like SequenceRenderer.render"""
    f32 = np.float32
    n_az = -(-int(n_points * oversample) // 64)
    rng = np.random.default_rng([seed, scene_id, frame])
    ex, ey, eth = path[frame, 0], path[frame, 1], path[frame, 2]
    ce, se = np.cos(eth, dtype=f32), np.sin(eth, dtype=f32)
    o = objs[objs[:, 5] > 0]
    wx = o[:, 0] + o[:, 6] * f32(frame * dt) - ex; wy = o[:, 1] + o[:, 7] * f32(frame * dt) - ey
    ox = ce * wx + se * wy; oy = -se * wx + ce * wy
    near = ox * ox + oy * oy < (MAX_RANGE + o[:, 3] + 4) ** 2
    o, ox, oy = o[near], ox[near], oy[near]
    beam = np.repeat(np.arange(64, dtype=f32), n_az); a = np.tile(np.arange(n_az, dtype=f32), 64)
    el = (f32(2.0) - beam * f32(26.8 / 63.0)) * f32(0.017453292)
    az = f32(-3.14159265) + (a + f32(rng.random())) * f32(6.2831853 / n_az)
    cel, sel = np.cos(el), np.sin(el)
    dx, dy, dz = cel * np.cos(az), cel * np.sin(az), sel
    denom = dx * f32(0.0174524) + dz * f32(0.9998477)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(denom < -1e-6, (-SENSOR_Z * f32(0.9998477)) / denom, f32(np.inf)).astype(f32)
    on_ground = np.ones(len(t), bool)
    ang = np.arctan2(dy, dx)
    for k in range(len(o)):
        # only the rays whose azimuth can meet the object (its bounding circle seen from the sensor)
        dist = float(np.hypot(ox[k], oy[k])); rad = float(np.hypot(o[k, 3], o[k, 4])) + 0.05
        if dist > rad:
            half = np.arcsin(min(1.0, rad / dist)); mid = np.arctan2(oy[k], ox[k])
            d = np.abs(((ang - mid + np.pi) % (2 * np.pi)) - np.pi)
            idx = np.nonzero(d <= half)[0]
        else:
            idx = np.arange(len(t))
        if not len(idx):
            continue
        c, s = np.cos(o[k, 2] - eth, dtype=f32), np.sin(o[k, 2] - eth, dtype=f32)
        px, py, pz = -(c * ox[k] + s * oy[k]), -(-s * ox[k] + c * oy[k]), SENSOR_Z - f32(0.5) * o[k, 5]
        rx, ry, rz = c * dx[idx] + s * dy[idx], -s * dx[idx] + c * dy[idx], dz[idx]
        with np.errstate(divide="ignore", invalid="ignore"):
            ix, iy, iz = 1 / rx, 1 / ry, 1 / rz
            hx, hy, hz = o[k, 3], o[k, 4], f32(0.5) * o[k, 5]
            t1, t2 = (-hx - px) * ix, (hx - px) * ix
            tmin, tmax = np.minimum(t1, t2), np.maximum(t1, t2)
            t1, t2 = (-hy - py) * iy, (hy - py) * iy
            tmin, tmax = np.maximum(tmin, np.minimum(t1, t2)), np.minimum(tmax, np.maximum(t1, t2))
            t1, t2 = (-hz - pz) * iz, (hz - pz) * iz
            tmin, tmax = np.maximum(tmin, np.minimum(t1, t2)), np.minimum(tmax, np.maximum(t1, t2))
        hit = (tmax >= tmin) & (tmin > 0.5) & (tmin < t[idx])
        t[idx[hit]] = tmin[hit]; on_ground[idx[hit]] = False
    with np.errstate(invalid="ignore"):
        lost = ~(t * cel < MAX_RANGE) | (rng.random(len(t)) < 0.01)
    gz = rng.normal(0, 1, len(t)).astype(f32) * np.where(on_ground, f32(0.02), f32(0.005))
    gx = rng.normal(0, 0.003, len(t)).astype(f32); gy = rng.normal(0, 0.003, len(t)).astype(f32)
    with np.errstate(invalid="ignore"):
        pts = np.stack([dx * t + gx, dy * t + gy, dz * t + gz, rng.random(len(t)).astype(f32)], 1).astype(f32)
    valid = ~lost
    c = np.cumsum(valid.astype(np.int64)); total = max(int(c[-1]), 1); nt = min(total, n_points)
    sel = valid & ((c * nt) // total > ((c - 1) * nt) // total)
    if order == "firing":
        perm = np.arange(64 * n_az).reshape(64, n_az).T.reshape(-1)
        pts, sel = pts[perm], sel[perm]
    elif order == "random":
        perm = rng.permutation(64 * n_az)
        pts, sel = pts[perm], sel[perm]
    return np.ascontiguousarray(pts[sel])


def render_sequence(scene_id: int, n_frames: int, n_points: int, scene: str = "street", order: str = "beam", density: float = 1.0):
    v, yaw = load_ego(n_frames)
    path = ego_path(v, yaw)
    objs = plaza_scene(scene_id, path) if scene == "plaza" else street_scene(scene_id, path, density=density)
    return [render_frame(objs, path, f, n_points, scene_id=scene_id, order=order) for f in range(n_frames)], v, yaw
