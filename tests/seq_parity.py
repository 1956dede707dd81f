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


NARROW = ("nonfinite", "yaw_rate", "covariance", "variance_sign")   # the filter has left the rails, by the reference's own kind of test
WIDE = NARROW + ("yaw_variance", "indefinite")                        # + a filter that is formally alive but numerically a random walk


def conditioning(state) -> tuple:
    """Why the CONTINUOUS state of a live track is not held to the 1e-4 bar on this frame: () = it is.

    NARROW reasons — what the reference itself treats as a diverged filter, two decades EARLIER than its own guards fire (it kills a track
    at P(4,4) > 1000 or det P > 10: OT/tracking/imm_ukf_jpda.cpp:826-851; by then the state has been noise for frames):
      nonfinite      NaN / Inf in x, P or the mode probabilities (a yaw variance blown up across +-pi: one frame before the guards)
      yaw_rate       |yaw rate| >= 20 rad/s
      covariance     a covariance entry >= 1e3
      variance_sign  a variance <= 0 (a coasting track's merged covariance soon stops being a covariance)
    WIDE reasons (round 3's criterion; kept ONLY as a label — a checker may opt into them with criterion="wide", and says so):
      yaw_variance   yaw or yaw-rate variance > 9 (sigma > 3 rad: the sigma points wrap around)
      indefinite     P has a non-positive eigenvalue
    A diverging filter amplifies last-bit differences of equivalent operation orders by decades per frame; discrete outputs (track set,
    trackManage, lifetime, flags) are compared on every track-frame regardless, and what is set aside is held against the measured
    noise floor of the reference's own arithmetic instead (NoiseFloor)."""
    x, P = np.asarray(state["x_merge"], np.float64), np.asarray(state["p_merge"], np.float64).reshape(5, 5)
    if not (np.isfinite(x).all() and np.isfinite(P).all() and np.isfinite(state["mode_prob"]).all()):
        return ("nonfinite",)
    r = []
    if abs(x[4]) >= 20.0:
        r.append("yaw_rate")
    if np.abs(P).max() >= 1e3:
        r.append("covariance")
    if np.diag(P).min() <= 0.0:
        r.append("variance_sign")
    if P[3, 3] > 9.0 or P[4, 4] > 9.0:
        r.append("yaw_variance")
    if not r and not np.linalg.eigvalsh((P + P.T) * 0.5).min() > 0.0:
        r.append("indefinite")
    return tuple(r)


def set_aside_reasons(state, criterion="narrow") -> tuple:
    allowed = NARROW if criterion == "narrow" else WIDE
    return tuple(k for k in conditioning(state) if k in allowed)


def well_conditioned(state, criterion="narrow") -> bool:
    return not set_aside_reasons(state, criterion)


TAINT_FRAMES = 30   # a track whose filter went through a diverging phase carries the amplified last-bit differences for a while after its
                    # covariance looks sane again (the measurements pull the state back within a few tens of frames)


def state_rel_err(sd, so):
    """max over the state keys of |difference| / max|reference| (per key, NaN entries aside) — the figure held to the 1e-4 bar;
    second value: False when the NaN patterns differ somewhere"""
    w, same_nan = 0.0, True
    for k in STATE_KEYS:
        so_k = np.asarray(so[k], np.float64).reshape(-1); sd_k = np.asarray(sd[k], np.float64).reshape(-1)
        nan = np.isnan(so_k)
        if not np.array_equal(nan, np.isnan(sd_k)):
            same_nan = False
            continue
        if nan.all():
            continue
        scale = max(float(np.abs(so_k[~nan]).max()), 1e-300)
        if scale > 1e-6:
            w = max(w, float(np.abs(sd_k[~nan] - so_k[~nan]).max()) / scale)
    return w, same_nan


class NoiseFloor:
    """The reference's OWN arithmetic noise, track-frame by track-frame: replicas of the reference tracker that differ from the oracle
    in floating-point operation order only, stepped with the same boxes —
      restatement   oracle/mot_oracle_track.c (plain C loops instead of Eigen expression templates)
      novec         the reference's sources rebuilt with -DEIGEN_DONT_VECTORIZE     (oracle/_ref/libmot_ref_novec.so)
      ref           the default reference build, when the primary oracle is the restatement (the tests)
    (whichever are on this box). All of these keep every fp32 expression of the reference as written and differ in the ORDER of fp64
    additions only. Deliberately NOT part of the floor: the build with FMA contraction (oracle/_ref/libmot_ref_fma.so, kinds=(..., "fma")).
    It also changes the reference's fp32 geometry (box centres, yaw choices) by an ulp, which flips decisions INSIDE the filter: it parts
    from the default build by up to 0.24 relative on perfectly conditioned track-frames and discretely within 11-153 frames
    (tests/test_tracker_noise_floor.py records it) — a floor that wide would explain anything. floor(i, so) = the largest state_rel_err of any
    replica against the primary oracle's state `so` of track i on the current frame: what "the same algorithm, rounded differently"
    amounts to for this track right now. A replica whose DISCRETE outputs part from the primary's (chaos reaching a gate decision) is
    retired from that frame on and reported."""

    def __init__(self, oracle, p, primary_is_ref: bool, instance: int = 0, kinds=("restatement", "ref", "novec")):
        """primary_is_ref: the oracle the device is compared with is oracle/_ref/libmot_ref.so itself (bench.py) — then the restatement
        is a replica; otherwise (the tests: oracle.Tracker is the primary) the default reference build is one.
        instance: which private copy of the reference builds to use (one per stream followed at the same time)."""
        self.reps, self.retired = {}, {}
        if primary_is_ref and "restatement" in kinds:
            # the C restatement ITSELF: when the checker was handed oracle_lib.RefFirst (the GPU suite), its Tracker() is another copy of the reference
            # build — identical to the primary, a replica that measures nothing (round 6: the measured conditioning found out)
            self.reps["restatement"] = getattr(oracle, "_b", oracle).Tracker(p)
        for k in [k for k in kinds if k not in ("restatement",) + (("ref",) if primary_is_ref else ())]:
            if p.seed_box_index == 1 and oracle.ref_variant(k, instance) is not None:   # (builds of package OT: preset 0 only)
                t = oracle.RefTracker(oracle.ref_variant(k, instance)); t.reset(); self.reps[k] = t
        self.frame = -1

    def names(self):
        return sorted(self.reps)

    def step(self, boxes_global, ts, ego_v, ego_yaw, primary_out, frame):
        self.frame = frame
        for k, t in list(self.reps.items()):
            t.ego_update(ts, ego_v, ego_yaw)
            o = t.step(boxes_global, ts, max_tracks=65536)
            if o["n"] != primary_out["n"] or any(not np.array_equal(o[q], primary_out[q]) for q in ("track_manage", "is_static", "is_vis")):
                self.retired[k] = frame
                if hasattr(t, "close"):
                    t.close()
                del self.reps[k]

    def floor(self, i, so):
        """None: no replica left to measure with (all retired / none on this box)"""
        if not self.reps:
            return None
        w = 0.0
        for t in self.reps.values():
            e, same = state_rel_err(t.state(i), so)
            w = max(w, e if same else float("inf"))
        return w

    def close(self):
        for t in self.reps.values():
            if hasattr(t, "close"):
                t.close()
        self.reps = {}


MEASURED_FLOOR = 1e-5   # the MEASURED conditioning (round-5 review, item 1b): a live track-frame is ill-conditioned iff the reference's own builds — replicas that differ
                        # from it in the order of fp64 additions only (NoiseFloor) — part by more than this on that very track-frame. On the complement the 1e-4 bar holds strictly.
FLOOR_FACTOR = 10.0   # a set-aside track-frame is EXPLAINED when the device's error is within this factor of the reference's own noise there


def note_conditioning(o, state_orc, frame, taint, criterion="narrow"):
    """call every frame: remembers until when a live track of the oracle is excluded from CONTINUOUS comparisons"""
    for i in np.nonzero(o["track_manage"] > 0)[0]:
        if not well_conditioned(state_orc(int(i)), criterion):
            taint[int(i)] = frame + TAINT_FRAMES


def compare_tracks(a, o, state_dev, state_orc, where, rtol=RTOL, stats=None, skip_ill_conditioned=False, taint=None, frame=None,
                   criterion="narrow", floor=None, assert_floor=False, measured=False, assert_measured=False):
    """a: the library's tracks of one stream (Context.get_tracks), o: the oracle's. Discrete outputs exact, continuous <= rtol.
    taint / frame: see note_conditioning (when given, this call also records the current conditioning).
    criterion: which conditioning reasons set a track-frame's continuous state aside ("narrow" | "wide").
    floor: NoiseFloor.floor — when given, every set-aside track-frame (and every one above the bar) is held against the reference's
    own noise there: `unexplained` counts those whose error exceeds both the bar and FLOOR_FACTOR x the floor (assert_floor: fail).
    measured (needs floor): the MEASURED conditioning — the floor is taken on EVERY live track-frame; one where the reference's own builds part by more than
    MEASURED_FLOOR (or differ in their NaN pattern, or no replica is left) is ill-conditioned, every other one is held to rtol STRICTLY, whatever the
    threshold criterion says about it (stats["measured"]; assert_measured: fail on the first violation)."""
    assert a["n"] == o["n"], (where, a["n"], o["n"])
    for k in ("track_manage", "is_static", "is_vis"):
        assert np.array_equal(a[k], o[k]), (where, k, np.nonzero(a[k] != o[k])[0][:8])
    if "lifetime" in o:
        assert np.array_equal(a["lifetime"], o["lifetime"]), (where, "lifetime")
    live = np.nonzero(o["track_manage"] > 0)[0]
    worst = 0.0
    for i in live:
        so = state_orc(int(i))
        why = set_aside_reasons(so, criterion)
        ill = bool(why)
        if taint is not None and frame is not None:
            if ill:
                taint[int(i)] = frame + TAINT_FRAMES
            elif taint.get(int(i), -1) >= frame:
                ill = True; why = ("recently_diverging",)
        fl = None
        m_ill = False
        if measured and floor is not None:   # the MEASURED conditioning decides what is held to the bar strictly
            fl = floor(int(i), so)
            m_ill = fl is None or not np.isfinite(fl) or fl > MEASURED_FLOOR
            ill = m_ill
            why = ("reference_builds_part",) if m_ill else ()
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
        w_key = None
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
                assert err <= rtol * scale + 1e-9, (where, int(i), k, err, scale, "SET ASIDE " + ",".join(why) if ill else "well conditioned",
                                                     "x_merge", list(np.asarray(so["x_merge"])), "diag P", list(np.diag(np.asarray(so["p_merge"]).reshape(5, 5))),
                                                     "lifetime", so["lifetime"], "track_manage", so["track_manage"])
            if scale > 1e-6:
                if err / scale > w_i:
                    w_key = (k, int(np.argmax(np.abs(np.where(nan, 0.0, sd_k - so_k)))))
                w_i = max(w_i, err / scale)
        if floor is not None and fl is None and not measured and (ill or w_i > RTOL):
            fl = floor(int(i), so)
        if measured and floor is not None:
            if stats is not None:
                m = stats.setdefault("measured", dict(well_conditioned=0, ill_conditioned=0, above_bar_well_conditioned=0, max_err_well_conditioned=0.0, above_bar_ill_conditioned=0,
                                                      max_err_over_floor_ill_conditioned=0.0, ill_without_replica=0, max_floor_well_conditioned=0.0))
                if m_ill:
                    m["ill_conditioned"] += 1
                    m["above_bar_ill_conditioned"] += int(w_i > RTOL)
                    if fl is None:
                        m["ill_without_replica"] += 1
                    elif np.isfinite(fl) and fl > 0 and w_i > RTOL:
                        m["max_err_over_floor_ill_conditioned"] = max(m["max_err_over_floor_ill_conditioned"], w_i / fl)
                else:
                    m["well_conditioned"] += 1
                    m["above_bar_well_conditioned"] += int(w_i > RTOL)
                    m["max_err_well_conditioned"] = max(m["max_err_well_conditioned"], w_i)
                    m["max_floor_well_conditioned"] = max(m["max_floor_well_conditioned"], fl)
            if assert_measured and not m_ill:
                assert w_i <= RTOL, (where, int(i), "error", w_i, "on a track-frame where the reference's own builds agree to", fl)
        if fl is not None:
            explained = w_i <= RTOL or w_i <= FLOOR_FACTOR * fl
            if assert_floor:
                assert explained, (where, int(i), "error", w_i, "noise floor of the reference's own arithmetic", fl, why)
        if stats is not None:
            key = "max_rel_state_err_ill_conditioned" if ill else "max_rel_state_err"
            stats[key] = max(stats.get(key, 0.0), w_i)
            if ill:
                stats["ill_conditioned"] = stats.get("ill_conditioned", 0) + 1
                for q in why:
                    stats.setdefault("set_aside_by", {})[q] = stats.setdefault("set_aside_by", {}).get(q, 0) + 1
            if w_i > RTOL:   # whatever the conditioning: how many live track-frames differ by more than the bar at all
                stats["above_bar"] = stats.get("above_bar", 0) + 1
                if fl is not None and fl > 0:
                    stats["above_bar_err_over_floor_max"] = max(stats.get("above_bar_err_over_floor_max", 0.0), w_i / fl)
                if len(stats.setdefault("above_bar_detail", [])) < 6 and w_key is not None:   # which entry of which matrix, and what the filter looked like
                    kk, at = w_key
                    stats["above_bar_detail"].append(dict(where=str(where), track=int(i), err=float(w_i), floor=None if fl is None else float(fl), set_aside=list(why), key=kk, entry=at,
                                                         device=float(np.asarray(sd[kk], np.float64).reshape(-1)[at]), oracle=float(np.asarray(so[kk], np.float64).reshape(-1)[at]),
                                                         lifetime=int(so["lifetime"])))
                if not ill:
                    stats["above_bar_well_conditioned"] = stats.get("above_bar_well_conditioned", 0) + 1
            if fl is not None:
                stats.setdefault("floors", []).append(fl)
                if ill:
                    stats.setdefault("set_aside_err_over_floor", []).append(w_i / fl if fl > 0 else (0.0 if w_i == 0 else float("inf")))
                if w_i > RTOL and w_i > FLOOR_FACTOR * fl:
                    stats["unexplained"] = stats.get("unexplained", 0) + 1
                    stats.setdefault("unexplained_detail", []).append(dict(where=str(where), track=int(i), err=w_i, floor=fl, why=list(why)))
                if ill and w_i > FLOOR_FACTOR * fl and w_i > RTOL * 1e-3:
                    stats["set_aside_above_10x_floor"] = stats.get("set_aside_above_10x_floor", 0) + 1
        if not ill:
            worst = max(worst, w_i)
    if stats is not None:
        stats.setdefault("max_rel_state_err", 0.0)
        stats["live_max"] = max(stats.get("live_max", 0), len(live))
        stats["tracks_ever"] = max(stats.get("tracks_ever", 0), int(o["n"]))
        stats["state_compares"] = stats.get("state_compares", 0) + len(live)
    return worst


def floor_summary(stats):
    """what the bench line / the tests report about the set-aside track-frames (JSON-able)"""
    fl = np.asarray(stats.get("floors", []), np.float64); fin = fl[np.isfinite(fl)]
    ratio = np.asarray(stats.get("set_aside_err_over_floor", []), np.float64); rfin = ratio[np.isfinite(ratio)]
    return {"track_frames_with_floor": int(len(fl)),
            "noise_floor": {"max": float(fin.max()) if len(fin) else None, "p50": float(np.median(fin)) if len(fin) else None,
                            "nan_pattern_differs_between_reference_builds": int(len(fl) - len(fin))},
            "device_err_over_floor": {"max": float(rfin.max()) if len(rfin) else None, "p50": float(np.median(rfin)) if len(rfin) else None},
            "set_aside_above_10x_floor": int(stats.get("set_aside_above_10x_floor", 0)),
            "above_1e-4_unexplained": int(stats.get("unexplained", 0)), "unexplained_detail": stats.get("unexplained_detail", [])[:8]}


def check_sequence(ctx, oracle, p, frame_ptr, host_frame, n_seq, stride, ego_v, ego_yaw, units, slots=None, rtol=RTOL,
                   check_labels=True, frames=None, skip_ill_conditioned=False, noise_floor=False, mar_check=False, measured=False):
    """Runs frames 0..F-1 of every slot through ctx.frames_dev and compares each frame of the slots in `slots` with the oracle.

    frame_ptr(f) -> device (or, on the emulator, host) address of frame f's batch [B][stride] float4
    host_frame(f, b) -> numpy (n, 4) copy of frame f of slot b
    n_seq[f][b] points, ego_v[f] / ego_yaw[f] ego motion (shared by the slots), units[b] timestamp step per slot (H11: 1e5 = the
    node's microsecond stamps, 0.1 = seconds). Returns statistics of what was compared.
    skip_ill_conditioned: a live track-frame the NARROW criterion sets aside (conditioning()) is not held to rtol; with noise_floor it
    is instead held to FLOOR_FACTOR x the reference's own arithmetic noise on that track-frame (NoiseFloor: replicas of the reference
    tracker per slot), asserted. No conditioning memory (taint) here: on these streams every track-frame outside the narrow criterion
    meets the bar (profiles/r04_tracker_noise_floor.jsonl).
    measured: the MEASURED conditioning instead of the threshold criterion (implies noise_floor): a live track-frame is ill-conditioned iff the
    reference's own builds part by more than MEASURED_FLOOR on it; EVERY other live track-frame is held to rtol strictly (asserted), the ill-conditioned
    ones to FLOOR_FACTOR x the floor measured there (asserted) — stats["measured"] carries the counts.
    mar_check: every cluster the restated box fit sends through the min-area-rectangle branch is cross-checked against the exhaustive
    integer oracle (tests/mar_check.py)."""
    if measured:
        noise_floor = skip_ill_conditioned = True
    F = len(n_seq) if frames is None else frames
    B = len(n_seq[0])
    slots = list(range(B)) if slots is None else list(slots)
    trackers = {b: oracle.Tracker(p) for b in slots}
    ref_primary = all(type(T).__name__ == "RefTracker" for T in trackers.values())   # oracle = oracle_lib.RefFirst on a box with oracle/_ref
    floors = {b: NoiseFloor(oracle, p, primary_is_ref=ref_primary, instance=100 + k) for k, b in enumerate(slots)} if noise_floor else {}
    stats = dict(frames=F, streams=len(slots), points=0, elevated=0, boxes=0, clusters=0, mar_clusters_cross_checked=0, mar_worst_area_err_units=0.0)
    if mar_check:
        import mar_check as MC
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
                if mar_check:
                    with oracle.observe_mar() as seen:
                        bx = oracle.box_fit(p, g["elevated"], cl["grid"], cl["num_cluster"])
                    for k, (pix, rect) in enumerate(seen):
                        try:
                            rel, _, clause = MC.check(oracle, pix, where=(where, "rectangle cluster", k))
                            if clause == "rounding":
                                stats["mar_worst_area_err_units"] = max(stats["mar_worst_area_err_units"], rel)
                            else:
                                stats["mar_thin_hulls_cosine_resolution"] = stats.get("mar_thin_hulls_cosine_resolution", 0) + 1
                        except AssertionError as e:   # collected, reported and (tests) asserted on by the caller; the point sets are kept for analysis
                            stats.setdefault("mar_failures", []).append(dict(where=str(where), cluster=k, what=str(e)[:300]))
                            stats.setdefault("_mar_failed_sets", []).append(pix)
                    stats["mar_clusters_cross_checked"] += len(seen)
                else:
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
                if noise_floor:
                    floors[b].step(gb, float(ts[b]), float(ego_v[f]), float(ego_yaw[f]), o, f)
                compare_tracks(at, o, lambda i: ctx.track_state(i, slot=b), T.state, where, rtol, stats, skip_ill_conditioned,
                               criterion="narrow", floor=floors[b].floor if noise_floor else None, assert_floor=noise_floor, measured=measured, assert_measured=measured)
                stats["points"] += n; stats["elevated"] += len(g["elevated"]); stats["boxes"] += len(bx["boxes"]); stats["clusters"] += cl["num_cluster"]
    finally:
        for T in trackers.values():
            T.close()
        for nf in floors.values():
            nf.close()
    stats["tracker_oracle"] = "reference build (oracle/_ref/libmot_ref.so)" if ref_primary else "restatement"
    if noise_floor:
        stats["noise_floor_replicas"] = {str(b): {"in_use": nf.names(), "retired_at_frame": nf.retired} for b, nf in floors.items()}
        stats.update(floor_summary(stats))
    failed = stats.pop("_mar_failed_sets", [])
    if failed:
        import os
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        np.savez_compressed(os.path.join(out, "mar_failed_sets_%d.npz" % os.getpid()), **{f"set{i}": a for i, a in enumerate(failed)})
    for k in ("floors", "set_aside_err_over_floor"):
        stats.pop(k, None)
    return stats
