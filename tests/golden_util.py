"""helpers shared by the golden-vector tests (tests/golden/*.npz, generated from oracle/_ref by make_golden.py)"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FRAMES = ("frame_ot_9k.npz", "frame_ot_24k.npz")
FRAMES_OT0 = ("frame_ot0_60k.npz",)   # the KITTI-tuned package (preset 1)
TRACKERS = ("tracker_ot_us.npz", "tracker_ot_sec.npz")
TRACKERS_OT0 = ("tracker_ot0_us.npz",)   # object_tracking0's tracker (preset 1)


def ego_of(fx, f):
    """ego speed and yaw fed to frame f when the fixture was generated"""
    return float(fx["ego_v"][f]), float(fx["ego_yaw"][f])


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def check_frame(fx, ground, cluster=None, boxes=None, polar=None):
    """ground: dict(elevated, ground); cluster: dict(grid, num_cluster); boxes: (nb,8,3)"""
    assert len(ground["elevated"]) == int(fx["n_elevated"]) and len(ground["ground"]) == int(fx["n_ground"])
    assert np.array_equal(ground["elevated"][:64, :3], fx["elevated_head"]) and np.array_equal(ground["ground"][:64, :3], fx["ground_head"])
    assert np.array_equal(ground["elevated"][:, :3].astype(np.float64).sum(0), fx["elevated_xyz_sum"])
    assert np.array_equal(ground["ground"][:, :3].astype(np.float64).sum(0), fx["ground_xyz_sum"])
    if polar is not None:
        assert np.array_equal(polar["min_z"], fx["min_z"]) and np.array_equal(polar["height"], fx["height"])
        assert np.array_equal(polar["is_ground"], fx["is_ground"])
        g = fx["is_ground"].astype(bool)
        assert np.array_equal(polar["hground"][g], fx["hground"][g])
    if cluster is not None:
        assert cluster["num_cluster"] == int(fx["num_cluster"])
        assert np.array_equal(cluster["grid"], fx["grid"].astype(np.int32))
    if boxes is not None:
        assert boxes.shape == fx["boxes"].shape and np.array_equal(boxes, fx["boxes"])


def check_tracker_frame(fx, f, out, state_fn, rtol=1e-4, atol=1e-9):
    """out: dict as returned by Tracker.step; state_fn(i) -> dict with x_merge, p_merge, mode_prob, lifetime"""
    n = int(fx["n_tracks"][f])
    assert out["n"] == n, (f, out["n"], n)
    assert np.array_equal(out["track_manage"], fx["track_manage"][f][:n]), f
    assert np.array_equal(out["is_static"], fx["is_static"][f][:n]) and np.array_equal(out["is_vis"], fx["is_vis"][f][:n]), f
    live = fx["track_manage"][f][:n] > 0
    assert np.allclose(out["p"][live], fx["pos"][f][:n][live], rtol=rtol, atol=1e-6), f
    assert np.allclose(out["v_yaw"][live], fx["v_yaw"][f][:n][live], rtol=rtol, atol=1e-7), f
    # dead tracks: the reference goes on reporting their frozen speed and (frozen yaw + the current ego yaw), imm_ukf_jpda.cpp:1012-1016 — so
    # does the library, for tracks still in a slot and for evicted ones (round 5; until then evicted tracks reported zeros)
    dead = ~live
    assert np.allclose(out["v_yaw"][dead], fx["v_yaw"][f][:n][dead], rtol=max(rtol, 1e-6), atol=1e-6, equal_nan=True), (f, "dead tracks' v / yaw")
    assert np.allclose(out["p"][dead][:, :2], fx["pos"][f][:n][dead][:, :2], rtol=max(rtol, 1e-6), atol=1e-6, equal_nan=True), (f, "dead tracks' position")
    for i in np.nonzero(live)[0]:
        s = state_fn(int(i))
        assert s["lifetime"] == fx["lifetime"][f][i], (f, i)
        for k in ("x_merge", "p_merge", "mode_prob"):
            ref = fx[k][f][i]
            scale = max(np.abs(ref).max(), 1e-300)
            assert np.abs(np.asarray(s[k]) - ref).max() <= rtol * scale + atol, (f, i, k, s[k], ref)
