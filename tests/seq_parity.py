"""End-to-end parity of a frame SEQUENCE through the fused device path against the oracle — shared by tests/test_sequence_gpu.py
(the MI355X, synth_dev-rendered 154-frame streams: BASELINE.json configs[3] as written), tests/test_emu_sequence.py (the same
checker on the emulator build, to keep its logic honest on CPU) and bench.py's `parity_check` (the benched frames themselves).

TEST INFRASTRUCTURE: imports the oracle (tests/oracle_lib.py). Per frame and stream, in the order the reference's nodes run:

  groundRemove          mask / elevated cloud / ground cloud              bit-exact   (OT/src/groundremove/ground_removal.cpp:177-249)
  componentClustering   label grid, cluster count, per-point labels       bit-exact   (OT/src/cluster/component_clustering.cpp:260-268)
  boxFitting            box corners and order                              bit-exact   (OT/src/cluster/box_fitting.cpp:422-435)
  tf step               boxes in the global frame                          bit-exact   (OT/tracking/main.cpp:76-83,143-158; oracle/ref_tf_capi.cpp)
  immUkfJpdaf           track count, trackManage, lifetime, static / vis   exact       (OT/tracking/imm_ukf_jpda.cpp:704-1112)
                        every state key of mot_track_state                 <= 1e-4 relative (BASELINE.json)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

RTOL = 1e-4
STATE_KEYS = ("x_merge", "x_cv", "x_ctrv", "x_rm", "p_merge", "p_cv", "p_ctrv", "p_rm", "mode_prob", "z_pred", "s", "k")


def bits_equal(a, b) -> bool:
    """bit-for-bit equality of two float32 arrays (-0 != +0, NaN == NaN with the same payload)"""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def apply_tf(m12, boxes):
    """pcl::transformPointCloud's arithmetic on a float 3 x 4 matrix: fp32, left to right"""
    m = np.asarray(m12, np.float32).reshape(3, 4); b = np.asarray(boxes, np.float32)
    x, y, z = b[..., 0], b[..., 1], b[..., 2]
    return np.stack([((m[r, 0] * x + m[r, 1] * y).astype(np.float32) + m[r, 2] * z).astype(np.float32) + m[r, 3] for r in range(3)], -1).astype(np.float32)


def boxes_to_global(oracle, lib, boxes, pose):
    """sensor -> global frame of the tracking node. With oracle/_ref present: the reference node's own call sequence
    (ref_boxes_to_global, oracle/ref_tf_capi.cpp); otherwise the float matrix of the library's host chain applied in numpy
    (tests/test_tf_exact.py pins that chain against the reference sequence and the golden fixture)."""
    boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 8, 3)
    if len(boxes) == 0:
        return boxes
    R = oracle.ref_tf()
    if R is not None:
        out = np.zeros_like(boxes)
        rc = R.ref_boxes_to_global(boxes.ctypes.data_as(C.c_void_p), len(boxes), C.c_double(pose[0]), C.c_double(pose[1]), C.c_double(pose[2]),
                                   out.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return out
    m = np.zeros(12, np.float32)
    assert lib.mot_debug_tf_matrix(C.c_double(pose[0]), C.c_double(pose[1]), C.c_double(pose[2]), m.ctypes.data_as(C.c_void_p)) == 0
    return apply_tf(m, boxes)


def well_conditioned(state) -> bool:
    """see tests/test_emu_tracker_random.py: a diverging filter amplifies last-bit differences by decades per frame until the
    reference's own guards kill the track; its discrete outputs are compared regardless"""
    x, P = np.asarray(state["x_merge"]), np.asarray(state["p_merge"])
    ok = np.isfinite(x).all() and np.isfinite(P).all() and np.isfinite(state["mode_prob"]).all()
    if not (ok and abs(x[4]) < 20.0 and np.abs(P).max() < 1e3 and np.diag(P.reshape(5, 5)).min() > 0.0):
        return False
    P = P.reshape(5, 5)
    if P[3, 3] > 9.0 or P[4, 4] > 9.0:   # the yaw (rate) is unknown to within a turn (sigma > 3 rad): the sigma points wrap around, the filter is a random walk
        return False
    return bool(np.linalg.eigvalsh((P + P.T) * 0.5).min() > 0.0)   # a covariance that is not positive definite: numerically meaningless


TAINT_FRAMES = 30   # a track whose filter went through a diverging phase carries the amplified last-bit differences for a while after its
                    # covariance looks sane again (the measurements pull the state back within a few tens of frames)


def note_conditioning(o, state_orc, frame, taint):
    """call every frame: remembers until when a live track of the oracle is excluded from CONTINUOUS comparisons"""
    for i in np.nonzero(o["track_manage"] > 0)[0]:
        if not well_conditioned(state_orc(int(i))):
            taint[int(i)] = frame + TAINT_FRAMES


def compare_tracks(a, o, state_dev, state_orc, where, rtol=RTOL, stats=None, skip_ill_conditioned=False, taint=None, frame=None):
    """a: the library's tracks of one stream (Context.get_tracks), o: the oracle's. Discrete outputs exact, continuous <= rtol.
    taint / frame: see note_conditioning (when given, this call also records the current conditioning)."""
    assert a["n"] == o["n"], (where, a["n"], o["n"])
    for k in ("track_manage", "is_static", "is_vis"):
        assert np.array_equal(a[k], o[k]), (where, k, np.nonzero(a[k] != o[k])[0][:8])
    if "lifetime" in o:
        assert np.array_equal(a["lifetime"], o["lifetime"]), (where, "lifetime")
    live = np.nonzero(o["track_manage"] > 0)[0]
    worst = 0.0
    for i in live:
        so = state_orc(int(i))
        ill = not well_conditioned(so)
        if taint is not None and frame is not None:
            if ill:
                taint[int(i)] = frame + TAINT_FRAMES
            ill = ill or taint.get(int(i), -1) >= frame
        check = not (ill and skip_ill_conditioned)   # discrete outputs were compared above regardless
        sd = state_dev(int(i))
        assert sd["lifetime"] == so["lifetime"] and sd["track_manage"] == so["track_manage"], (where, int(i))
        # (a track of the reference can go NaN — a yaw variance blown up across +-pi — one frame before its guards kill it: NaN
        # must then be NaN on both sides, in the same entries)
        if check:   # the outputs, relative to the size of the vector they belong to (a velocity of 1e-3 m/s next to a yaw of 3 rad is not a scale)
            for key, atol in (("p", 1e-6), ("v_yaw", 1e-7), ("vis_box", 1e-5)):
                av, ov = np.asarray(a[key][i], np.float64), np.asarray(o[key][i], np.float64)
                nan = np.isnan(ov)
                assert np.array_equal(nan, np.isnan(av)), (where, int(i), key, "NaN pattern")
                if not nan.all() and np.isfinite(rtol):   # (rtol = inf: the caller only collects the state errors)
                    assert np.abs(av[~nan] - ov[~nan]).max() <= rtol * np.abs(ov[~nan]).max() + atol, (where, int(i), key, av, ov)
        w_i = 0.0
        for k in STATE_KEYS:
            so_k = np.asarray(so[k], np.float64); sd_k = np.asarray(sd[k], np.float64).reshape(so_k.shape)
            nan = np.isnan(so_k)
            if check:
                assert np.array_equal(nan, np.isnan(sd_k)), (where, int(i), k, "NaN pattern")
            if nan.all() or not np.array_equal(nan, np.isnan(sd_k)):
                continue
            scale = max(float(np.abs(so_k[~nan]).max()), 1e-300)
            err = float(np.abs(sd_k[~nan] - so_k[~nan]).max())
            if check and np.isfinite(rtol):
                assert err <= rtol * scale + 1e-9, (where, int(i), k, err, scale, "ILL-CONDITIONED" if ill else "well conditioned",
                                                     "x_merge", list(np.asarray(so["x_merge"])), "diag P", list(np.diag(np.asarray(so["p_merge"]).reshape(5, 5))),
                                                     "lifetime", so["lifetime"], "track_manage", so["track_manage"])
            if scale > 1e-6:
                w_i = max(w_i, err / scale)
        if stats is not None:
            key = "max_rel_state_err_ill_conditioned" if ill else "max_rel_state_err"
            stats[key] = max(stats.get(key, 0.0), w_i)
            if ill:
                stats["ill_conditioned"] = stats.get("ill_conditioned", 0) + 1
            if w_i > RTOL:   # whatever the conditioning: how many live track-frames differ by more than the bar at all
                stats["above_bar"] = stats.get("above_bar", 0) + 1
                if not ill:
                    stats["above_bar_well_conditioned"] = stats.get("above_bar_well_conditioned", 0) + 1
        if not ill:
            worst = max(worst, w_i)
    if stats is not None:
        stats.setdefault("max_rel_state_err", 0.0)
        stats["live_max"] = max(stats.get("live_max", 0), len(live))
        stats["tracks_ever"] = max(stats.get("tracks_ever", 0), int(o["n"]))
        stats["state_compares"] = stats.get("state_compares", 0) + len(live)
    return worst


def check_sequence(ctx, oracle, p, frame_ptr, host_frame, n_seq, stride, ego_v, ego_yaw, units, slots=None, rtol=RTOL,
                   check_labels=True, frames=None, skip_ill_conditioned=False):
    """Runs frames 0..F-1 of every slot through ctx.frames_dev and compares each frame of the slots in `slots` with the oracle.

    frame_ptr(f) -> device (or, on the emulator, host) address of frame f's batch [B][stride] float4
    host_frame(f, b) -> numpy (n, 4) copy of frame f of slot b
    n_seq[f][b] points, ego_v[f] / ego_yaw[f] ego motion (shared by the slots), units[b] timestamp step per slot (H11: 1e5 = the
    node's microsecond stamps, 0.1 = seconds). Returns statistics of what was compared."""
    F = len(n_seq) if frames is None else frames
    B = len(n_seq[0])
    slots = list(range(B)) if slots is None else list(slots)
    trackers = {b: oracle.Tracker(p) for b in slots}
    taints = {}
    stats = dict(frames=F, streams=len(slots), points=0, elevated=0, boxes=0, clusters=0)
    try:
        for f in range(F):
            ts = np.array([1.0e9 + f * units[b] for b in range(B)], np.float64)
            ctx.frames_dev(frame_ptr(f), stride * 4, n_seq[f], run_tracker=True, timestamps=ts, ego_v=np.full(B, ego_v[f]), ego_yaw=np.full(B, ego_yaw[f]))
            for b in slots:
                n = int(n_seq[f][b]); where = (f, b)
                cloud = np.ascontiguousarray(host_frame(f, b)[:n])
                g = oracle.ground_remove(p, cloud)
                a = ctx.get_ground(b, n_hint=n)
                assert np.array_equal(a["mask"], g["mask"]), (where, "mask", int((a["mask"] != g["mask"]).sum()))
                assert bits_equal(a["elevated"], g["elevated"]) and bits_equal(a["ground"], g["ground"]), (where, "clouds")
                cl = oracle.cluster(p, g["elevated"])
                ac = ctx.get_clusters(b, n_elevated=len(g["elevated"]) if check_labels else 0)
                assert ac["num_cluster"] == cl["num_cluster"] and np.array_equal(ac["grid"], cl["grid"]), (where, "label grid")
                if check_labels:
                    assert np.array_equal(ac["point_label"], cl["point_label"]), (where, "point labels")
                bx = oracle.box_fit(p, g["elevated"], cl["grid"], cl["num_cluster"])
                ab = ctx.get_boxes(b)
                assert bits_equal(ab["boxes"], bx["boxes"]), (where, "boxes", len(ab["boxes"]), len(bx["boxes"]))
                assert np.array_equal(ab["box_cluster"], bx["box_cluster"]) and ab["n_undefined"] == bx["n_undefined"], (where, "box clusters")
                T = trackers[b]
                ego = T.ego_update(float(ts[b]), float(ego_v[f]), float(ego_yaw[f]))
                gb = boxes_to_global(oracle, ctx.lib, bx["boxes"], ego[:3])
                if len(gb):   # what track_prep_kernel handed the tracker: the node's tf step, bit for bit
                    gdev = np.zeros((1024, 8, 3), np.float32)
                    assert ctx.lib.mot_debug_copy(ctx._h, 11, b, gdev.ctypes.data_as(C.c_void_p), C.c_size_t(gdev.nbytes)) == 0
                    assert bits_equal(gdev[: len(gb)], gb), (where, "boxes in the global frame")
                o = T.step(gb, float(ts[b]), max_tracks=max(ctx.max_tracks_total, 64))
                at = ctx.get_tracks(b)
                compare_tracks(at, o, lambda i: ctx.track_state(i, slot=b), T.state, where, rtol, stats, skip_ill_conditioned,
                               taint=taints.setdefault(b, {}), frame=f)
                stats["points"] += n; stats["elevated"] += len(g["elevated"]); stats["boxes"] += len(bx["boxes"]); stats["clusters"] += cl["num_cluster"]
    finally:
        for T in trackers.values():
            T.close()
    return stats
