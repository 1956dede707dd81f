"""The tracker's 1e-4 bar on the streams the bench and the GPU sequence tests run, with the exemption turned into evidence (round-3
review, item 1). Fixture tests/golden/track_boxes.npz: the boxes (global frame) the tracker is fed on six rendered streams — bench streams
0-3, the seconds-stamped stream, the 200 k-point stream (tools/dump_track_streams.py on the MI355X + tools/make_track_box_fixture.py; the
boxes are bit-equal to the reference's own, tests/test_sequence_gpu.py).

1. The reference's OWN arithmetic noise: oracle/_ref/libmot_ref.so (primary) against replicas that differ in the order of fp64 additions
   only (the C restatement; the reference's sources rebuilt without Eigen's packet kernels) — same boxes, every frame. They follow the
   primary discretely through every frame; on track-frames no criterion sets aside they agree to <= 1e-5; on the set-aside ones (NARROW
   criterion: NaN, |yaw rate| >= 20, a covariance entry >= 1e3, a non-positive variance — at most 2 % of the live track-frames of a
   microsecond-stamped stream) they part by up to 2e-2: that is the floor no implementation can be held under.
   Recorded next to it, NOT used as a floor: the same sources with FMA contraction (-ffp-contract=fast -mfma). That build changes the
   reference's fp32 geometry by an ulp, and the reference's filter is knife-edged enough for that to flip decisions inside it: up to 0.24
   relative on well-conditioned track-frames, a different track set within 11-153 frames. The 1e-4 bar is tighter than the reference's own
   reproducibility across compiler flags.
2. The device code (emulator build: the kernels' operation order with the host's libm — the MI355X itself runs the same check in
   tests/test_sequence_gpu.py and bench.py's parity_check) on the same boxes: EVERY live track-frame within 1e-4 of the reference, or set
   aside by the narrow criterion AND within 10 x the floor measured there. Discrete outputs equal on every frame.
Reference: OT/tracking/imm_ukf_jpda.cpp:704-1112 (its own divergence guards :826-851), ukf.cpp:630-902."""
import os
import sys

import numpy as np
import pytest

import seq_parity as SP

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "track_boxes.npz")


def streams():
    d = np.load(FIX)
    for name in d["streams"]:
        name = str(name)
        scene, points, unit, preset, F = d[name + "/meta"]
        off = np.concatenate([[0], np.cumsum(d[name + "/n_boxes"])])
        bx = d[name + "/boxes_global"]
        yield name, float(unit), int(preset), [bx[off[f]:off[f + 1]] for f in range(int(F))], d[name + "/ego_v"], d[name + "/ego_yaw"]


def _need_ref(oracle):
    if oracle.ref() is None:
        pytest.skip("oracle/_ref is not on this box")


def test_reference_arithmetic_noise_floor(oracle):
    _need_ref(oracle)
    p = oracle.params(0)
    report = {}
    for name, unit, preset, boxes, ego_v, ego_yaw in streams():
        R = oracle.RefTracker(); R.reset()
        NF = SP.NoiseFloor(oracle, p, primary_is_ref=True)
        assert {"restatement", "novec"} <= set(NF.names())
        FM = SP.NoiseFloor(oracle, p, primary_is_ref=True, kinds=("fma",))   # observed, not part of the floor
        n_live = n_aside = 0
        worst_ok, worst_aside, worst_fma = 0.0, 0.0, 0.0
        for f, gb in enumerate(boxes):
            ts = 1.0e9 + f * unit
            R.ego_update(ts, float(ego_v[f]), float(ego_yaw[f]))
            o = R.step(gb, ts, max_tracks=65536)
            NF.step(gb, ts, float(ego_v[f]), float(ego_yaw[f]), o, f); FM.step(gb, ts, float(ego_v[f]), float(ego_yaw[f]), o, f)
            for i in np.nonzero(o["track_manage"] > 0)[0]:
                so = R.state(int(i)); fl = NF.floor(int(i), so)
                n_live += 1
                if SP.set_aside_reasons(so, "narrow"):
                    n_aside += 1; worst_aside = max(worst_aside, fl if np.isfinite(fl) else 1.0)
                else:
                    worst_ok = max(worst_ok, fl)
                    if SP.well_conditioned(so, "wide") and FM.names():
                        worst_fma = max(worst_fma, FM.floor(int(i), so))
        # the restatement and the unvectorised rebuild follow the reference DISCRETELY through all 154 frames of every stream
        assert "restatement" not in NF.retired and "novec" not in NF.retired, (name, NF.retired)
        NF.close(); FM.close()
        report[name] = dict(live=n_live, set_aside=n_aside, floor_well_conditioned=worst_ok, floor_set_aside=worst_aside,
                            fma_build_parts_discretely_at_frame=FM.retired.get("fma"), fma_build_max_err_on_well_conditioned=worst_fma)
        assert worst_ok <= 1e-5, (name, report[name])                      # equivalent arithmetic agrees where the filter is sane ...
        if unit > 1:
            assert n_aside <= 0.02 * n_live, (name, report[name])          # ... and what is set aside is <= 2 % of a microsecond-stamped stream
    assert max(r["floor_set_aside"] for r in report.values()) > 1e-3        # ... where the reference's own builds part by more than the bar
    print("reference arithmetic noise floor:", report)


@pytest.mark.parametrize("perturb", [False, True])
def test_emulated_device_within_noise_floor(mot, oracle, perturb):
    """perturb: the kernels' sin / cos / exp / atan2 / pow answer one ulp off for two arguments in three (what a device math library may do)"""
    _need_ref(oracle)
    if perturb:
        os.environ["MOT_EMU_PERTURB"] = "1"
    try:
        import importlib
        import build_emu
        importlib.reload(build_emu)
        lib = build_emu.build()
    finally:
        os.environ.pop("MOT_EMU_PERTURB", None)
        import build_emu as _b
        importlib.reload(_b)
    p = oracle.params(0)
    for name, unit, preset, boxes, ego_v, ego_yaw in streams():   # all six streams, both modes
        R = oracle.RefTracker(); R.reset()
        NF = SP.NoiseFloor(oracle, p, primary_is_ref=True)
        stats = {}
        with mot.Context(lib_path=lib, max_points=1024, max_tracks_total=256) as c:
            for f, gb in enumerate(boxes):
                ts = 1.0e9 + f * unit
                R.ego_update(ts, float(ego_v[f]), float(ego_yaw[f])); c.ego_update(ts, float(ego_v[f]), float(ego_yaw[f]))
                o = R.step(gb, ts, max_tracks=65536); a = c.track_step(gb, ts)
                NF.step(gb, ts, float(ego_v[f]), float(ego_yaw[f]), o, f)
                o["lifetime"] = a["lifetime"]   # (the reference API has no lifetime vector: compared per live track through the states)
                SP.compare_tracks(a, o, c.track_state, R.state, (name, f), stats=stats, skip_ill_conditioned=True, floor=NF.floor, assert_floor=True)
        NF.close()
        assert stats.get("above_bar_well_conditioned", 0) == 0 and stats.get("unexplained", 0) == 0, (name, stats)
        assert stats["max_rel_state_err"] <= SP.RTOL


PLAZA_FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "track_boxes_plaza.npz")


def test_dense_scene_measured_conditioning(mot, oracle):
    """configs[3]'s "<= 64 tracks" on CPU: the boxes the tracker is fed on the bench's dense (plaza) scene 7000 — dumped on the MI355X
    (tools/dump_track_streams.py, MOT_DUMP_SCENE_KIND=plaza; bit-equal to the reference's own boxes there: tests/test_sequence_gpu.py) — replayed through the
    reference build, its two replicas and the device code on the emulator. The MEASURED conditioning (round-5 review, item 1b): a live track-frame is
    ill-conditioned iff the replicas part from the reference by more than 1e-5 on it; on EVERY other one (>= 99.5 % of 9 309) the device code is within
    1e-4 — asserted — and on this stream by a factor of ten; the 38 ill-conditioned ones carry every above-1e-4 difference (31 of them), within 3 x the reference's
    own spread. profiles/r06_tracker_parity_study.md: the same count with sequential sums (30) and with the libm an ulp off (32) — it is chaos, not the device."""
    _need_ref(oracle)
    import build_emu
    lib = build_emu.build()
    d = np.load(PLAZA_FIX)
    name = str(d["streams"][0])
    off = np.concatenate([[0], np.cumsum(d[name + "/n_boxes"])]); bx = d[name + "/boxes_global"]
    ego_v, ego_yaw = d[name + "/ego_v"], d[name + "/ego_yaw"]
    p = oracle.params(0)
    R = oracle.RefTracker(); R.reset()
    NF = SP.NoiseFloor(oracle, p, primary_is_ref=True)
    assert {"restatement", "novec"} <= set(NF.names())
    stats = {}
    with mot.Context(lib_path=lib, max_points=1024, max_tracks_total=1024) as c:
        for f in range(len(ego_v)):
            gb = bx[off[f]:off[f + 1]]; ts = 1.0e9 + f * 1e5
            R.ego_update(ts, float(ego_v[f]), float(ego_yaw[f])); c.ego_update(ts, float(ego_v[f]), float(ego_yaw[f]))
            o = R.step(gb, ts, max_tracks=65536); a = c.track_step(gb, ts)
            NF.step(gb, ts, float(ego_v[f]), float(ego_yaw[f]), o, f)
            o["lifetime"] = a["lifetime"]
            SP.compare_tracks(a, o, c.track_state, R.state, (name, f), stats=stats, skip_ill_conditioned=True, floor=NF.floor, assert_floor=True, measured=True, assert_measured=True)
    assert not NF.retired, NF.retired
    NF.close()
    m = stats["measured"]
    assert stats["live_max"] >= 60 and m["well_conditioned"] + m["ill_conditioned"] == stats["state_compares"] > 9000
    assert m["above_bar_well_conditioned"] == 0 and m["max_err_well_conditioned"] <= 1e-5 and m["ill_without_replica"] == 0, m
    assert m["ill_conditioned"] <= 0.005 * stats["state_compares"] and m["above_bar_ill_conditioned"] > 0, m   # the chaos is there, and it is rare
    print("dense scene, measured conditioning:", m)
