"""Randomised box sequences through the tracker kernel (emulated) against the restated tracker: objects that move, stop, vanish
and reappear, split into two boxes (the over-segmentation merge), crowd each other (shared gates), plus clutter boxes and a
wandering ego pose. Discrete outputs (track set, management states, static / shown flags, lifetimes) must be equal, the
continuous state within 1e-6 relative. MOT_PROP_SCALE multiplies the number of sequences for long exploration runs."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
SCALE = int(os.environ.get("MOT_PROP_SCALE", "1"))


def well_conditioned(state):
    """continuous states are compared only for tracks whose filter has not started to diverge: an ill-conditioned track
    (covariance blown up, turn rate of hundreds of rad/s, NaN) amplifies last-bit differences of equivalent operation orders
    by many decades per frame before the reference's own guards kill it; the discrete outputs are compared regardless"""
    x, P = np.asarray(state["x_merge"]), np.asarray(state["p_merge"])
    ok = np.isfinite(x).all() and np.isfinite(P).all() and np.isfinite(state["mode_prob"]).all()
    # a coasting track's merged covariance soon stops being positive definite (negative variances): numerically meaningless
    return bool(ok and abs(x[4]) < 20.0 and np.abs(P).max() < 1e3 and np.diag(P.reshape(5, 5)).min() > 0.0)


def box(cx, cy, w, l, yaw, top):
    c, s = np.cos(yaw), np.sin(yaw)
    corners = np.array([[-l / 2, -w / 2], [l / 2, -w / 2], [l / 2, w / 2], [-l / 2, w / 2]])
    xy = corners @ np.array([[c, s], [-s, c]]) + [cx, cy]
    b = np.zeros((8, 3), np.float32)
    b[:4, :2] = xy; b[4:, :2] = xy; b[:4, 2] = -2.0; b[4:, 2] = top
    return b


def sequence(seed, frames=14):
    rng = np.random.default_rng(seed)
    n_obj = int(rng.integers(1, 7))
    pos = rng.uniform(-25, 25, (n_obj, 2)); vel = rng.uniform(-1.2, 1.2, (n_obj, 2)) * (rng.random((n_obj, 1)) < 0.7)
    size = rng.uniform(0.4, 5.0, (n_obj, 2)); yaw = rng.uniform(-np.pi, np.pi, n_obj); top = rng.uniform(-1.2, 0.6, n_obj)
    if n_obj > 1 and rng.random() < 0.5:
        pos[1] = pos[0] + rng.uniform(-1.5, 1.5, 2)     # two objects inside each other's gates
    out = []
    for f in range(frames):
        boxes = []
        for o in range(n_obj):
            if rng.random() < 0.12:
                continue                                  # missed detection
            p = pos[o] + vel[o] * f * 0.1 * rng.uniform(5, 12) + rng.normal(0, 0.03, 2)
            y = yaw[o] + rng.normal(0, 0.02) + (np.pi / 2 if rng.random() < 0.1 else 0.0)   # the L-shape fit flips by 90 degrees now and then
            s = size[o] * rng.uniform(0.85, 1.15, 2)
            if rng.random() < 0.15:                       # over-segmentation: the object arrives as two boxes
                d = np.array([np.cos(y), np.sin(y)]) * s[1] / 4
                boxes += [box(*(p - d), s[0], s[1] / 2, y, top[o]), box(*(p + d), s[0], s[1] / 2, y, top[o])]
            else:
                boxes.append(box(p[0], p[1], s[0], s[1], y, top[o]))
        for _ in range(int(rng.integers(0, 3)) if rng.random() < 0.4 else 0):
            boxes.append(box(*rng.uniform(-28, 28, 2), *rng.uniform(0.3, 3, 2), rng.uniform(-3, 3), rng.uniform(-1, 0.5)))   # clutter
        rng.shuffle(boxes)
        out.append((np.array(boxes, np.float32).reshape(-1, 8, 3), 1.0e9 + f * 1.0e5, 2.0 + 0.3 * np.sin(f * 0.7 + seed), 0.01 * f * ((seed % 3) - 1)))
    return out


def hostile_sequence(seed, frames=16):
    """four steady objects plus, per frame, one degenerate input: an exact duplicate measurement, a zero-area box, a box
    10 km away, an empty frame, every box three times, a near-duplicate 0.1 mm off"""
    rng = np.random.default_rng(seed); out = []
    base = [box(rng.uniform(-20, 20), rng.uniform(-20, 20), rng.uniform(0.5, 3), rng.uniform(0.5, 5), rng.uniform(-3, 3), 0.3) for _ in range(4)]
    for f in range(frames):
        bs = [b.copy() for b in base]
        for b in bs:
            b[:, :2] += f * 0.2
        kind = int(rng.integers(0, 6))
        if kind == 0: bs.append(bs[0].copy())
        if kind == 1: z = bs[1].copy(); z[:, :2] = z[0, :2]; bs.append(z)
        if kind == 2: h = bs[2].copy(); h[:, :2] *= 1e4; bs.append(h)
        if kind == 3: bs = []
        if kind == 4: bs = bs * 3
        if kind == 5: t = bs[3].copy(); t[:, :2] += 1e-4; bs.append(t)
        out.append((np.array(bs, np.float32).reshape(-1, 8, 3), 1e9 + f * 1e5, 1.0, 0.0))
    return out


@pytest.mark.parametrize("preset", [0, 1])
def test_emu_tracker_degenerate_measurements(mot, oracle, preset):
    import build_emu
    lib = build_emu.build()
    p = oracle.params(preset)
    with mot.Context(mot.params(preset, lib=mot.load_library(lib)), lib_path=lib, max_points=4096, max_tracks_total=2048) as c:
        for seed in range(8 * SCALE):
            c.reset(); T = oracle.Tracker(p)
            for f, (boxes, ts, v, yaw) in enumerate(hostile_sequence(seed)):
                c.ego_update(ts, v, yaw); T.ego_update(ts, v, yaw)
                a = c.track_step(boxes, ts); o = T.step(boxes, ts)
                assert a["n"] == o["n"], (seed, f)
                for k in ("track_manage", "is_static", "is_vis", "lifetime", "vis_box"):
                    assert np.array_equal(a[k], o[k]), (seed, f, k)
            T.close()


@pytest.mark.parametrize("preset", [0, 1])
def test_emu_tracker_random_sequences(mot, oracle, preset):
    import build_emu
    lib = build_emu.build()
    p = oracle.params(preset)
    n_seq = 12 * SCALE
    with mot.Context(mot.params(preset, lib=mot.load_library(lib)), lib_path=lib, max_points=4096, max_tracks_total=512) as c:
        for seed in range(1000 * preset, 1000 * preset + n_seq):
            c.reset()
            T = oracle.Tracker(p)
            for f, (boxes, ts, v, yaw) in enumerate(sequence(seed)):
                assert np.allclose(c.ego_update(ts, v, yaw), T.ego_update(ts, v, yaw), rtol=1e-12, atol=1e-12)
                a = c.track_step(boxes, ts); o = T.step(boxes, ts)
                assert a["n"] == o["n"], (seed, f, a["n"], o["n"])
                for k in ("track_manage", "is_static", "is_vis", "lifetime"):
                    assert np.array_equal(a[k], o[k]), (seed, f, k, a[k], o[k])
                assert np.array_equal(a["vis_box"], o["vis_box"]), (seed, f)
                for i in np.nonzero(o["track_manage"] > 0)[0]:
                    sa, so = c.track_state(int(i)), T.state(int(i))
                    if not well_conditioned(so):
                        continue
                    assert np.allclose(a["p"][i], o["p"][i], rtol=1e-5, atol=1e-6) and np.allclose(a["v_yaw"][i], o["v_yaw"][i], rtol=1e-6, atol=1e-7), (seed, f, int(i))
                    for k in ("x_merge", "p_merge", "mode_prob"):
                        scale = max(np.abs(so[k]).max(), 1e-300)
                        assert np.abs(np.asarray(sa[k]) - so[k]).max() <= 1e-6 * scale + 1e-9, (seed, f, int(i), k)
            T.close()
