"""tests/seq_parity.py (the end-to-end sequence checker of tests/test_sequence_gpu.py and bench.py's parity_check) exercised on
the emulator build of the kernels — a check of the CHECKER's logic and of the fused path's host side on CPU, not a parity claim
(tests/emu/hipemu.h). Two ragged streams with different timestamp units, a moving ego, every frame compared."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


import pytest


@pytest.mark.parametrize("which", ["restatement", "reference build first", "reference build first, measured conditioning"])
def test_sequence_checker_on_the_emulator(mot, synth, oracle, which):
    """which: the oracle the checker is given — the restatement, or (as on the GPU box) oracle_lib.RefFirst; "measured": the conditioning of
    tests/test_sequence_gpu.py::test_154_frame_dense_scene_measured_conditioning (seq_parity.MEASURED_FLOOR)"""
    measured = which.endswith("measured conditioning")
    import build_emu
    import seq_parity as SP
    lib = build_emu.build()
    if which != "restatement":
        if oracle.ref() is None:
            pytest.skip("oracle/_ref is not on this box")
        oracle = oracle.RefFirst(oracle)
    B, N, stride, F = 2, 6000, 6144, 8
    clouds = np.zeros((F, B, stride, 4), np.float32)
    n_seq = np.zeros((F, B), np.int32)
    for f in range(F):
        for b in range(B):
            n = N - 500 * b - 7 * f
            clouds[f, b, :n] = synth.make_cloud(N, 20 + b, f)[:n]
            n_seq[f, b] = n
    ego_v = 2.0 + 0.2 * np.arange(F); ego_yaw = 0.01 * np.arange(F)
    p = oracle.params(0)
    with mot.Context(lib_path=lib, max_points=stride, max_batch=B, max_tracks_total=256) as c:
        st = SP.check_sequence(c, oracle, p, lambda f: clouds[f].ctypes.data, lambda f, b: clouds[f, b], n_seq, stride, ego_v, ego_yaw, units=[1e5, 0.1],
                               skip_ill_conditioned=True, noise_floor=True, mar_check=True, measured=measured)   # the GPU run's settings (tests/seq_parity_gpu_run.py)
    if measured:
        m = st["measured"]
        assert m["well_conditioned"] + m["ill_conditioned"] == st["state_compares"] and m["above_bar_well_conditioned"] == 0 and m["ill_without_replica"] == 0, m
    assert st["frames"] == F and st["boxes"] > 0 and st["tracks_ever"] > 0 and st["max_rel_state_err"] <= SP.RTOL
    assert st["above_1e-4_unexplained"] == 0 and st["mar_clusters_cross_checked"] > 0
    assert st["tracker_oracle"].startswith("restatement" if which == "restatement" else "reference build")
    if which != "restatement":
        assert ("box_fit", "reference build (box -> cluster ids: restatement)") in oracle.used and ("Tracker", "reference build") in oracle.used


def test_sequence_mode_equals_frame_by_frame(mot, synth):
    """mot_sequence_dev (K consecutive frames of ONE stream in one call: stateless stages as one batch, the tracker chained on the device)
    against K calls of mot_frames_dev with one frame each — every stage's results of every frame and the tracker's outputs and full
    filter states after every frame, bit for bit (the same kernels on the same inputs in the same order). Emulator build."""
    import ctypes as C
    import build_emu
    import seq_parity as SP
    lib = build_emu.build()
    K, N, stride = 9, 7000, 7168
    clouds = np.zeros((K, stride, 4), np.float32); n = np.zeros(K, np.int32)
    for f in range(K):
        n[f] = N - 13 * f; clouds[f, : n[f]] = synth.make_cloud(N, 31, f)[: n[f]]
    ts = 1.0e9 + 1.0e5 * np.arange(K); ev = 2.0 + 0.1 * np.arange(K); ey = 0.004 * np.arange(K)
    per_frame = []
    with mot.Context(lib_path=lib, max_points=stride, max_batch=1, max_tracks_total=128) as c:
        for f in range(K):
            c.frames_dev(clouds[f].ctypes.data, stride * 4, [n[f]], run_tracker=True, timestamps=[ts[f]], ego_v=[ev[f]], ego_yaw=[ey[f]])
            tr = c.get_tracks(0)
            gb = np.zeros((1024, 8, 3), np.float32); assert c.lib.mot_debug_copy(c._h, 11, 0, gb.ctypes.data_as(C.c_void_p), C.c_size_t(gb.nbytes)) == 0
            per_frame.append(dict(boxes=c.get_boxes(0)["boxes"], grid=c.get_clusters(0)["grid"], ne=c.get_ground(0, want_clouds=False)["n_elevated"], tracks=tr,
                                  gb=gb[: len(c.get_boxes(0)["boxes"])].copy(),
                                  states={int(i): c.track_state(int(i)) for i in np.nonzero(tr["track_manage"] > 0)[0]}))
    rec = np.dtype([("id", "i4"), ("track_manage", "i4"), ("is_static", "i4"), ("is_vis", "i4"), ("p", "f4", 3), ("lifetime", "i4"), ("v_yaw", "f8", 2), ("vis_box", "f4", 24)])
    with mot.Context(lib_path=lib, max_points=stride, max_batch=K, max_tracks_total=128) as c:
        out = np.zeros((K, 64), rec); cnt = np.full(K, -1, np.int32)
        c.sequence_dev(clouds.ctypes.data, stride * 4, n, ts, ev, ey, out.ctypes.data, 64, cnt.ctypes.data)
        c.synchronize()
        for f in range(K):
            r = per_frame[f]
            assert SP.bits_equal(c.get_boxes(f)["boxes"], r["boxes"]) and np.array_equal(c.get_clusters(f)["grid"], r["grid"]) and c.get_ground(f, want_clouds=False)["n_elevated"] == r["ne"], f
            gb = np.zeros((1024, 8, 3), np.float32); assert c.lib.mot_debug_copy(c._h, 11, f, gb.ctypes.data_as(C.c_void_p), C.c_size_t(gb.nbytes)) == 0
            assert SP.bits_equal(gb[: len(r["gb"])], r["gb"]), (f, "boxes in the global frame")
            live = np.nonzero(r["tracks"]["track_manage"] > 0)[0]   # the per-frame record block: the live tracks after frame f, in id order
            assert cnt[f] == len(live) and np.array_equal(out[f]["id"][: cnt[f]], live), (f, cnt[f], live)
            for key in ("track_manage", "is_static", "is_vis", "lifetime"):
                assert np.array_equal(out[f][key][: cnt[f]], r["tracks"][key][live]), (f, key)
            bits = lambda a: np.ascontiguousarray(a).view(np.uint8)
            assert np.array_equal(bits(out[f]["p"][: cnt[f]]), bits(r["tracks"]["p"][live])) and np.array_equal(bits(out[f]["v_yaw"][: cnt[f]]), bits(r["tracks"]["v_yaw"][live]))
        last = per_frame[-1]; tr = c.get_tracks(0)
        assert tr["n"] == last["tracks"]["n"] and all(np.array_equal(tr[k], last["tracks"][k]) for k in ("track_manage", "lifetime", "is_static", "is_vis", "p", "v_yaw", "vis_box"))
        for i, so in last["states"].items():
            sd = c.track_state(i)
            for k in SP.STATE_KEYS:
                assert np.array_equal(np.asarray(sd[k]), np.asarray(so[k])), (i, k)
        assert tr["n"] > 3 and len(last["states"]) > 1
        # the by-products of frame k, on demand (mot_get_ground re-runs the compaction of the batch = the sequence; labels from cells and grid):
        # what the stage-wise calls return for the same frame
        for f in (0, K // 2, K - 1):
            with mot.Context(lib_path=lib, max_points=stride) as st:
                g = st.ground_remove(clouds[f, : n[f]]); cl = st.cluster(g["elevated"])
            got = c.get_ground(f, n_hint=int(n[f]))
            assert np.array_equal(got["elevated"], g["elevated"]) and np.array_equal(got["ground"], g["ground"]) and np.array_equal(got["mask"][: n[f]], g["mask"]), f
            assert np.array_equal(c.get_clusters(f, len(g["elevated"]))["point_label"], cl["point_label"]), f
        # a second sequence continues the first (the stream's state carries over): frames K.. of the same drive
        with pytest.raises(mot.MotError):
            c.sequence_dev(clouds.ctypes.data, stride * 4, np.r_[n, n[:1]], np.r_[ts, ts[:1]], np.r_[ev, ev[:1]], np.r_[ey, ey[:1]])   # more frames than slots


def test_sequence_mode_with_empty_single_point_and_full_frames(mot, synth):
    """a drive whose frames are ragged to the extreme — 0 points, 1 point, a full slot, 64 points — in sequence mode against the frame-by-frame loop"""
    import build_emu
    lib = build_emu.build()
    stride = 7168; ns = [7000, 0, 1, 6999, 64, 7168, 3000]; K = len(ns)
    clouds = np.zeros((K, stride, 4), np.float32)
    for f, n in enumerate(ns):
        clouds[f, :n] = synth.make_cloud(7168, 31, f)[:n]
    ts = 1e9 + 1e5 * np.arange(K); ev = 2.0 + 0.1 * np.arange(K); ey = 0.004 * np.arange(K)
    ref = []
    with mot.Context(lib_path=lib, max_points=stride, max_batch=1, max_tracks_total=64) as c:
        for f in range(K):
            c.frames_dev(clouds[f].ctypes.data, stride * 4, [ns[f]], run_tracker=True, timestamps=[ts[f]], ego_v=[ev[f]], ego_yaw=[ey[f]])
            ref.append((c.get_boxes(0)["boxes"].copy(), c.get_tracks(0)))
    with mot.Context(lib_path=lib, max_points=stride, max_batch=K, max_tracks_total=64) as c:
        c.sequence_dev(clouds.ctypes.data, stride * 4, np.array(ns, np.int32), ts, ev, ey); c.synchronize()
        for f in range(K):
            assert np.array_equal(c.get_boxes(f)["boxes"].view(np.uint32), ref[f][0].view(np.uint32)), f
        tr = c.get_tracks(0)
        for k in ("track_manage", "lifetime", "p", "v_yaw"):
            assert np.array_equal(tr[k].view(np.uint8), ref[-1][1][k].view(np.uint8)), k
        assert tr["n"] >= 2 and sum(len(r[0]) for r in ref) >= 4
