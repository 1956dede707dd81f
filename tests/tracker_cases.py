"""Tracker load cases shared by the emulator test (tests/test_emu_tracker_load.py, CPU) and the MI355X test
(tests/test_tracker_gpu.py): bodies only; the callers supply the library and how a numpy array becomes a device buffer.
TEST INFRASTRUCTURE (compares with the oracle)."""
import numpy as np

import seq_parity as SP


def grid_boxes(streams, T, f, vel, rng, spacing):
    """`T` boxes per stream on a sqrt(T) x sqrt(T) lattice `spacing` metres apart, drifting with `vel` (bench.py tracker_stress's
    generator): spacing 9 m keeps every gate to itself; 2 m makes neighbouring tracks share gated boxes — the cross-track
    matchingVec bookkeeping of imm_ukf_jpda.cpp:232 at the size BASELINE.json configs[3] names (<= 64 tracks)"""
    side = int(np.ceil(np.sqrt(T)))
    centres = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2)[:T] * spacing - side * spacing / 2
    ctr = centres[None] + vel * (0.1 * f) + rng.normal(0, 0.02, size=(streams, T, 2))
    bx = np.zeros((streams, T, 8, 3), np.float32)
    w, l = (3.8, 1.7) if spacing >= 6 else (1.2, 0.8)
    bx[..., :2] = ctr[:, :, None, :] + np.array([[0, 0], [w, 0], [w, l], [0, l]] * 2)[None, None]
    bx[:, :, :4, 2] = -2.0; bx[:, :, 4:, 2] = -0.4
    return bx


def many_live_tracks(mot, oracle, to_dev, lib_path=None, streams=3, T=64, frames=30, spacing=9.0, seed=11, min_live=64):
    """mot_track_steps_dev (every stream at once, boxes on the device) against the oracle with `T` simultaneously live tracks
    per stream: every frame, every stream, discrete outputs exact and every state key <= 1e-4"""
    rng = np.random.default_rng(seed)
    vel = rng.uniform(-1.0, 1.0, size=(streams, T, 2))
    p = oracle.params(0)
    kw = dict(lib_path=lib_path) if lib_path else {}
    stats = {}
    with mot.Context(max_points=1024, max_batch=streams, max_tracks_total=1024, **kw) as c:
        Ts = [oracle.Tracker(p) for _ in range(streams)]
        stride = T * 24
        live = [0] * streams
        for f in range(frames):
            ts = 1.0e9 + f * 1e5
            bx = grid_boxes(streams, T, f, vel, rng, spacing)
            ptr, free = to_dev(bx.reshape(streams, stride))
            for s in range(streams):
                a_ego = c.ego_update(ts, 0.0, 0.0, s); o_ego = Ts[s].ego_update(ts, 0.0, 0.0)
                assert np.array_equal(a_ego, o_ego)
            c.track_steps_dev(ptr, stride, [T] * streams, [ts] * streams)
            for s in range(streams):
                a = c.get_tracks(s); o = Ts[s].step(bx[s], ts, max_tracks=1024)
                SP.compare_tracks(a, o, lambda i: c.track_state(i, slot=s), Ts[s].state, (f, s), stats=stats, skip_ill_conditioned=spacing < 6)
                live[s] = int((o["track_manage"] > 0).sum())
            free()
        for T_ in Ts:
            T_.close()
    assert min(live) >= min_live, live
    stats["live_last"] = live
    return stats


def angle_far_beyond_32_turns(mot, oracle, lib_path=None):
    """The one documented deviation from the reference (csrc/track.hip wrap_pi): up to 32 turns the normalisation loop runs as
    written, beyond it whole turns come off in one step. Two ways past 32 turns: (a) an ego yaw of hundreds of radians — the
    output yaw is wrap(x_merge(3) + egoYaw), OT/tracking/imm_ukf_jpda.cpp:1010 — and (b) a time step of thousands of seconds,
    which carries every CTRV sigma point yaw + yawd * dt round and round (ukf.cpp:539-571). The oracle keeps the reference's
    loops (oracle/mot_oracle_track.c). Discrete outputs must stay equal on every frame; the yaw outputs agree to 1e-9."""
    import test_emu_tracker_random as TR
    p = oracle.params(0)
    kw = dict(lib_path=lib_path) if lib_path else {}
    rng = np.random.default_rng(5)
    n_obj = 6
    pos = rng.uniform(-20, 20, (n_obj, 2)); vel = rng.uniform(-1.5, 1.5, (n_obj, 2)); yaw = rng.uniform(-3, 3, n_obj)
    hit_ego = hit_dt = False
    with mot.Context(max_points=1024, max_tracks_total=512, **kw) as c:
        T = oracle.Tracker(p)
        ts = 1.0e9
        for f in range(34):
            jump = f in (16, 25)
            ts += 6.0e9 if jump else 1.0e5                      # (b): dt = 6000 s on two frames
            ego_yaw = 0.01 * f + (900.0 if f >= 8 else 0.0)     # (a): 143 turns from frame 8 on
            boxes = np.array([TR.box(*(pos[o] + vel[o] * 0.1 * f), 1.8, 4.2, yaw[o] + 0.02 * f, -0.3) for o in range(n_obj)], np.float32)
            a_ego = c.ego_update(ts, 3.0, ego_yaw); o_ego = T.ego_update(ts, 3.0, ego_yaw)
            assert np.allclose(a_ego, o_ego, rtol=1e-12, atol=1e-9)
            pre = [T.state(i) for i in range(c.get_tracks(0)["n"])] if jump else []   # states the 6000 s prediction starts from
            a = c.track_step(boxes, ts); o = T.step(boxes, ts)
            assert a["n"] == o["n"], f
            for k in ("track_manage", "is_static", "is_vis", "lifetime"):
                assert np.array_equal(a[k], o[k]), (f, k)
            livei = np.nonzero(o["track_manage"] > 0)[0]
            fin = np.isfinite(o["v_yaw"][livei, 1])
            assert np.allclose(a["v_yaw"][livei, 1][fin], o["v_yaw"][livei, 1][fin], rtol=0, atol=1e-9), f
            if f >= 8 and len(livei):
                hit_ego = True
            if jump:
                hit_dt |= any(s["track_manage"] > 0 and abs(s["x_ctrv"][4]) * 6000.0 > 64 * np.pi for s in pre)
        T.close()
    assert hit_ego and hit_dt


def blinking_world(seed, spots, frames):
    """objects that appear at fixed spots of a lattice, drift a little, vanish, and come back: the tracks of a spot die and new ones
    are born where dead ones lie — over a long run far more tracks are created than are ever alive, and a new track's visible box
    regularly contains the last position of a dead one (the reference's merge step looks at those too, imm_ukf_jpda.cpp:666-700)"""
    import test_emu_tracker_random as TR
    rng = np.random.default_rng(seed)
    side = int(np.ceil(np.sqrt(spots)))
    centre = (np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2)[:spots] - side / 2) * 7.0 + rng.uniform(-1, 1, (spots, 2))
    state = np.zeros(spots, np.int64)           # > 0: frames the object stays; < 0: frames until it is back
    state[:] = rng.integers(1, 40, spots) * np.where(rng.random(spots) < 0.7, 1, -1)
    drift = rng.uniform(-0.6, 0.6, (spots, 2)); size = rng.uniform(0.8, 4.5, (spots, 2)); yaw = rng.uniform(-3, 3, spots)
    age = np.zeros(spots)
    for f in range(frames):
        boxes = []
        for k in range(spots):
            if k % 3 == 0:
                # Every third spot replays, with a period of 200 frames, the one situation in which a DEAD track changes a live one's fate in
                # the reference (mergeOverSegmentation runs over every track ever created, imm_ukf_jpda.cpp:666-700): a small object D lives
                # and dies at (2, 0.9); later a large static object A stands over the spot and a long object B drives into A's box from the
                # right. B's centre inside A's visible box would merge B away (A is older) — unless B's own box holds some track's position:
                # D's last position does that job for a while. An implementation that forgets dead tracks kills B the moment it enters.
                t = (f + 37 * k) % 200
                c0 = centre[k] + rng.normal(0, 0.01, 2)
                if t < 25:
                    boxes.append(TR.box(c0[0] + 2.0, c0[1] + 0.9, 1.0, 1.0, 0.0, -0.3))
                if 40 <= t < 190:
                    boxes.append(TR.box(c0[0], c0[1], 3.0, 6.0, 0.0, -0.3))                                # A: x in [-3, 3], y in [-1.5, 1.5]
                if 42 <= t < 190:
                    boxes.append(TR.box(c0[0] + 8.0 - 0.06 * (t - 42), c0[1] + 0.9, 1.0, 3.0, 0.0, -0.3))   # B: 3 m long, 1 m wide, moving left
                continue
            if state[k] > 0:
                p = centre[k] + drift[k] * 0.1 * age[k] + rng.normal(0, 0.03, 2)
                boxes.append(TR.box(p[0], p[1], size[k, 0], size[k, 1], yaw[k] + rng.normal(0, 0.02), -0.3))
                age[k] += 1; state[k] -= 1
                if state[k] == 0:
                    state[k] = -int(rng.integers(4, 30)); age[k] = 0; drift[k] = rng.uniform(-0.6, 0.6, 2)
            else:
                state[k] += 1
                if state[k] == 0:
                    state[k] = int(rng.integers(12, 70))
        if rng.random() < 0.2:
            boxes.append(TR.box(*rng.uniform(-30, 30, 2), *rng.uniform(0.5, 2.5, 2), rng.uniform(-3, 3), -0.2))   # clutter
        yield np.array(boxes, np.float32).reshape(-1, 8, 3), 1.0e9 + f * 1.0e5, 1.0 + 0.5 * np.sin(0.01 * f), 0.0005 * f


def long_run_bounded_slots(mot, oracle, lib_path=None, frames=1500, slots=16, spots=10, seed=3, state_every=25, min_ever_factor=4, max_chaos_restarts=0,
                           ref_frames=500):
    """SURVEY.md H14 / the reference never frees a track: a long run on `slots` track slots must give what the oracle gives with
    unbounded memory — every frame the discrete outputs of EVERY track ever created (reference index order), every `state_every`
    frames the filter states of the live ones — while far more tracks are created than there are slots.

    max_chaos_restarts (the MI355X run; 0 on the emulator, whose arithmetic is the oracle's up to operation order): the reference's
    filter diverges now and then (a covariance that stops being positive definite; its own guards kill the track a few frames
    later). While it lasts, the track's state is numerical noise — last-bit differences of the device's sin / cos / exp grow to
    O(1) within three frames (the same run on the emulator with those functions perturbed by one ulp, MOT_EMU_PERTURB, parts
    from the oracle at exactly the same frame) — and when such a track's gate decides about a box, the noise becomes a discrete
    difference (a birth more or less). That is the reference's chaos, not a property of the implementation: a discrete mismatch
    is accepted ONLY while the oracle has a live track that is or was ill-conditioned within the last 30 frames, both sides are
    then started over, and the number of such restarts is bounded."""
    if hasattr(oracle, "restatement"):
        # The GPU suite's oracle is the reference build first (oracle_lib.RefFirst) — not here: the reference's tracker never frees a track and
        # walks all of them every frame, so thousands of frames with thousands of tracks ever created cost minutes of host time (5 min for the
        # two long runs on the GPU box, measured in round 4) where the restatement takes seconds; and this world is chaos by construction —
        # two builds of the REFERENCE part discretely within a few hundred frames of it (tests/test_tracker_noise_floor.py: the floor's
        # replicas are retired at frame 45-399 of a blinking world), so "equal to libmot_ref.so for 10 000 frames" is not a property any
        # second implementation, or any second compilation of the first, has. The restatement is pinned to libmot_ref.so on CPU
        # (tests/test_oracle_vs_ref.py), and every other tracker test of the GPU suite runs against libmot_ref.so itself.
        oracle = oracle.restatement("Tracker", "long chaotic run: the reference's tracker is O(tracks ever) per frame, and its own rebuilds part discretely in this world")
    p = oracle.params(0)
    kw = dict(lib_path=lib_path) if lib_path else {}
    stats = {"chaos_restarts": 0, "frames_compared": 0}
    taint = {}
    ever_total = 0
    # THE REFERENCE'S OWN BUILD for the head of the run (round-4 review: these runs never met the reference's code on the GPU box): for the
    # first `ref_frames` frames — while the tracks ever created are few and its O(tracks ever) step is cheap — oracle/_ref/libmot_ref.so
    # (and its -DEIGEN_DONT_VECTORIZE rebuild) are stepped beside the restatement. Their DISCRETE outputs must equal the restatement's (a
    # replica that parts from it is chaos reaching a gate decision: it is retired and the frame recorded), and their state differences are
    # the noise floor the device's states are held against under the NARROW criterion on those frames (assert: every live track-frame
    # within 1e-4, or within 10 x the reference's own noise there). Beyond ref_frames: the wide criterion at 1e-2, see below.
    floor = None
    if ref_frames and getattr(oracle, "ref", lambda: None)() is not None:
        floor = SP.NoiseFloor(oracle, p, primary_is_ref=False, instance=3, kinds=("ref", "novec"))
        stats["reference_builds_stepped"] = floor.names()
    with mot.Context(max_points=1024, max_tracks_total=slots, **kw) as c:
        T = oracle.Tracker(p)
        o = None
        for f, (boxes, ts, v, yaw) in enumerate(blinking_world(seed, spots, frames)):
            assert np.allclose(c.ego_update(ts, v, yaw), T.ego_update(ts, v, yaw), rtol=1e-12, atol=1e-12)
            a = c.track_step(boxes, ts); o_prev = o; o = T.step(boxes, ts, max_tracks=1 << 16)
            assert not a["capacity_exceeded"], f
            if floor is not None:
                if f < ref_frames and floor.reps:
                    floor.step(boxes, ts, v, yaw, o, f)
                    stats["reference_frames"] = f + 1
                else:
                    stats["reference_builds_retired_at"] = dict(floor.retired); floor.close(); floor = None
            equal = a["n"] == o["n"] and all(np.array_equal(a[k], o[k]) for k in ("track_manage", "is_static", "is_vis", "lifetime"))
            if not equal:
                chaotic = any(until >= f - 1 for until in taint.values())
                assert chaotic and stats["chaos_restarts"] < max_chaos_restarts, (f, a["n"], o["n"], "discrete outputs differ", "oracle has a diverging track" if chaotic else "NO diverging track")
                stats["chaos_restarts"] += 1; ever_total += o["n"]
                c.reset(); T.reset(); taint.clear(); o = None
                continue
            stats["frames_compared"] += 1
            live = o["track_manage"] > 0
            dead = ~live
            assert np.array_equal(a["vis_box"][dead & (o["is_vis"] == 0)], o["vis_box"][dead & (o["is_vis"] == 0)])
            SP.note_conditioning(o, T.state, f, taint, criterion="wide")
            if f % state_every == 0 or f == frames - 1:
                # continuous values: filter states of the live tracks (a track that is or recently was diverging is compared in its discrete
                # outputs only: seq_parity.well_conditioned / note_conditioning), and the last positions the dead tracks left behind (the
                # merge step keeps reading them)
                # (rtol 1e-2 here, not the 1e-4 of the sequence tests: this world is built to stress the track BOOKKEEPING — overlapping
                # boxes, tracks driven into each other for thousands of frames — and keeps long-lived filters loose (yaw-rate variances
                # of 5-6 (rad/s)^2) or lets them pass through short indefinite phases; such a filter carries the device's last-bit
                # differences at the 1e-3 level long after its covariance looks sane again. The discrete outputs of every track ever
                # created, compared exactly on every frame, are what this test is about.)
                # criterion "wide" + the 30-frame conditioning memory: THIS world needs them, the rendered streams do not. With the narrow
                # criterion the MI355X run of round 4 (gpurun session r4s1) failed here at frame 2900 on a track with yaw / yaw-rate variances
                # of 11.3 / 10.4 (rad)^2 — a filter that is alive by every test of the reference and a random walk in yaw — by 1.7e-2 in
                # p_merge. The counts per criterion are in stats["set_aside_by"].
                SP.compare_tracks(a, o, c.track_state, T.state, f, rtol=1e-2, stats=stats, skip_ill_conditioned=True, taint=taint, frame=f, criterion="wide")
            if floor is not None and floor.reps and (f % 5 == 0):   # the head of the run, against the reference's own builds: narrow criterion + noise floor
                head = stats.setdefault("head", {})
                # (assert_floor: a track-frame above 1e-4 — set aside or not — fails unless the reference's own builds differ about as much there.
                # On the MI355X this world does produce a few well-conditioned track-frames at 2-3e-4: loose filters, where libmot_ref.so and
                # its -DEIGEN_DONT_VECTORIZE rebuild are 1.7-1.9e-4 apart themselves — counted in head["above_bar_well_conditioned"], reported.)
                SP.compare_tracks(a, o, c.track_state, T.state, f, rtol=float("inf"), stats=head, criterion="narrow", floor=floor.floor, assert_floor=True)
                for i in np.nonzero(dead)[0][-64:]:
                    if SP.well_conditioned(T.state(int(i)), "wide") and taint.get(int(i), -1) < 0:
                        assert np.allclose(a["p"][i][:2], o["p"][i][:2], rtol=1e-2, atol=1e-4), (f, int(i), "position of a dead track")
                        # ... and its frozen speed / (frozen yaw + the current ego yaw), which the reference keeps reporting (imm_ukf_jpda.cpp:1012-1016)
                        dv = np.abs(a["v_yaw"][i] - o["v_yaw"][i]); dv[1] = min(dv[1], abs(2 * np.pi - dv[1]))   # (a yaw at +-pi may wrap either way)
                        assert np.all(dv <= 1e-2 * np.maximum(np.abs(o["v_yaw"][i]), 1.0)), (f, int(i), "v / yaw of a dead track", a["v_yaw"][i], o["v_yaw"][i])
            stats["live_peak"] = max(stats.get("live_peak", 0), int(live.sum()))
        ever_total += o["n"] if o is not None else 0
        T.close()
        if floor is not None:
            stats["reference_builds_retired_at"] = dict(floor.retired); floor.close()
    if "head" in stats:
        h = stats.pop("head")
        stats["head_vs_reference_builds"] = {k: h.get(k) for k in ("state_compares", "max_rel_state_err", "above_bar", "above_bar_well_conditioned", "ill_conditioned", "unexplained") if k in h}
        stats["head_vs_reference_builds"]["noise_floor_max"] = max([x for x in h.get("floors", []) if np.isfinite(x)] + [0.0])
    assert ever_total >= min_ever_factor * slots and stats["live_peak"] <= slots, (ever_total, stats)
    stats["tracks_ever"] = ever_total
    return stats
