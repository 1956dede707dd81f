"""Error convention of the C-ABI (int status + mot_last_error, never abort), exercised on the emulator build:
argument checks and capacity limits are host code shared with the GPU build."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.fixture(scope="module")
def emu(mot):
    import build_emu
    lib = build_emu.build()
    return lib, mot.load_library(lib)


def test_create_rejects_bad_arguments(mot, emu):
    lib, L = emu
    h = C.c_void_p()
    p = mot.params(0, lib=L)
    assert L.mot_create(None, 0, 1024, 1, 16, C.byref(h)) == mot.MOT_E_ARG
    assert L.mot_create(C.byref(p), 0, 0, 1, 16, C.byref(h)) == mot.MOT_E_ARG          # no capacity
    assert L.mot_create(C.byref(p), 0, 1024, 0, 16, C.byref(h)) == mot.MOT_E_ARG       # no stream slot
    assert L.mot_create(C.byref(p), 7, 1024, 1, 16, C.byref(h)) == mot.MOT_E_ARG       # device ordinal out of range
    for field, value in (("num_grid", 4), ("num_grid", 100000), ("gauss_samples", 5), ("ram_points", 0), ("pic_scale", 1000.0)):
        q = mot.params(0, lib=L, **{field: value})
        assert L.mot_create(C.byref(q), 0, 1024, 1, 16, C.byref(h)) == mot.MOT_E_ARG, field
    assert not h.value


def test_calls_report_capacity_and_argument_errors(mot, emu, synth):
    lib, L = emu
    with mot.Context(lib_path=lib, max_points=2048, max_batch=2, max_tracks_total=4) as c:
        big = synth.make_cloud(5000, 1, 0)
        with pytest.raises(mot.MotError) as e:
            c.ground_remove(big)
        assert e.value.code == mot.MOT_E_CAPACITY and "max_points" in str(e.value)
        with pytest.raises(mot.MotError) as e:
            c.cluster(big)
        assert e.value.code == mot.MOT_E_CAPACITY
        with pytest.raises(mot.MotError) as e:
            c.ground_remove_pointcloud2(np.zeros(16 * 10, np.uint8), 10, 16, 0, 4, 14)     # z field sticks out of the record
        assert e.value.code == mot.MOT_E_ARG
        with pytest.raises(mot.MotError) as e:
            c.track_state(0)                                                              # no such track yet
        assert e.value.code == mot.MOT_E_ARG
        with pytest.raises(mot.MotError) as e:
            c.frames_dev(big.ctypes.data, 2048 * 4, [100, 100, 100])                      # more frames than slots
        assert e.value.code == mot.MOT_E_ARG
        n1 = np.array([100], np.int32)   # tracker without stamps / ego motion (the Python wrapper always passes them: call the ABI)
        assert L.mot_frames_dev(c._h, C.c_void_p(big.ctypes.data), C.c_long(2048 * 4), n1.ctypes.data_as(C.c_void_p), 1, 1, None, None, None) == mot.MOT_E_ARG
        assert b"run_tracker needs" in L.mot_last_error(c._h)
        # more tracks than provisioned: the reference never frees a track, so births beyond max_tracks_total are a capacity error
        boxes = np.zeros((12, 8, 3), np.float32)
        for k in range(12):
            boxes[k, :, :2] = np.array([[0, 0], [2, 0], [2, 1], [0, 1]] * 2) + [6.0 * k - 30, 5.0]
            boxes[k, 4:, 2] = 1.0; boxes[k, :4, 2] = -2.0
        full = False   # (MOT_E_CAPACITY from the C call with the records delivered; the binding turns it into `capacity_exceeded`)
        for f in range(6):
            ts = 1.0e9 + f * 1e5
            c.ego_update(ts, 0.0, 0.0); full |= c.track_step(boxes, ts)["capacity_exceeded"]
        assert full
        # a frame with more boxes than the library takes is REFUSED before the step: n_tracks = -1, the stream's tracks untouched (round-5 advice: the
        # adapter used to read it as "births dropped" and wiped the stream)
        many = np.zeros((mot.MOT_MAX_BOXES_PER_FRAME + 1, 8, 3), np.float32)
        arr = (mot.MotTrack * 64)(); nt = C.c_int(7)
        assert L.mot_track_step(c._h, 0, many.ctypes.data_as(C.c_void_p), len(many), C.c_double(2.0e9), arr, 64, C.byref(nt)) == mot.MOT_E_CAPACITY and nt.value == -1
        with pytest.raises(mot.MotError) as e:
            c.track_step(many, 2.0e9)
        assert e.value.code == mot.MOT_E_CAPACITY and "not taken" in str(e.value)
        # the context stays usable after an error
        c.reset()
        small = synth.make_cloud(1500, 1, 0)
        assert len(c.ground_remove(small)["mask"]) == 1500


def test_track_records_never_outgrow_a_buffer_of_max_tracks_ever(mot, emu):
    """the contract a long-running consumer (ros/src/mot_ros_common.hpp) relies on: with mot_params.max_tracks_ever = the size of its
    record buffer, n_tracks never exceeds the buffer however many tracks the world would create — births beyond the budget are dropped,
    MOT_E_CAPACITY (sticky) comes WITH the records, and mot_reset_tracks_slot clears it"""
    import tracker_cases as TC
    lib, L = emu
    cap = 10
    with mot.Context(mot.params(0, lib=L, max_tracks_ever=cap), lib_path=lib, max_points=1024, max_tracks_total=cap) as c:
        arr = (mot.MotTrack * cap)(); nt = C.c_int(0)
        rcs, restarted = [], 0
        for f, (boxes, ts, v, yaw) in enumerate(TC.blinking_world(5, 9, 200)):
            c.ego_update(ts, v, yaw)
            b = np.ascontiguousarray(boxes, np.float32)
            rc = L.mot_track_step(c._h, 0, b.ctypes.data_as(C.c_void_p), len(b), C.c_double(ts), arr, cap, C.byref(nt))
            assert rc in (mot.MOT_OK, mot.MOT_E_CAPACITY), (f, rc, L.mot_last_error(c._h))
            assert 0 <= nt.value <= cap, (f, nt.value)                       # the records were delivered
            if nt.value:
                assert [arr[i].id for i in range(nt.value)] == list(range(nt.value))
            rcs.append(rc)
            if rc == mot.MOT_E_CAPACITY:
                if restarted == 0:   # sticky until the tracks are started over
                    assert L.mot_get_tracks(c._h, 0, arr, cap, C.byref(nt)) == mot.MOT_E_CAPACITY and nt.value <= cap
                assert L.mot_reset_tracks_slot(c._h, 0) == mot.MOT_OK
                restarted += 1
        assert restarted >= 2 and rcs[0] == mot.MOT_OK


def test_round_4_entry_points_check_their_arguments(mot, emu, synth):
    """mot_box_markers, mot_stream_*, mot_sequence_dev, mot_set_tracker_mode: null pointers, slots out of range, undersized buffers — a status
    and a message, the context stays usable"""
    lib, L = emu
    with mot.Context(lib_path=lib, max_points=8192, max_batch=2, max_tracks_total=32) as c:
        h = c._h
        n = C.c_int(0); six = (C.c_float * 60)()
        assert L.mot_box_markers(h, 0, six, 10, None) == mot.MOT_E_ARG          # null n_boxes
        assert L.mot_box_markers(h, 2, six, 10, C.byref(n)) == mot.MOT_E_ARG     # no such slot
        assert L.mot_box_markers(h, 0, six, -1, C.byref(n)) == mot.MOT_E_ARG
        assert L.mot_box_markers(h, 0, None, 0, C.byref(n)) == mot.MOT_E_STATE   # nothing fitted yet
        cloud = synth.make_cloud(8000, 2, 0)
        g = c.ground_remove(cloud); c.cluster(g["elevated"]); b = c.box_fit_resident()
        if len(b["boxes"]) > 1:
            assert L.mot_box_markers(h, 0, six, 1, C.byref(n)) == mot.MOT_E_CAPACITY and n.value == len(b["boxes"])
        assert L.mot_box_markers(h, 0, None, 1024, C.byref(n)) == mot.MOT_OK and n.value == len(b["boxes"])   # count only
        # a stage-wise call that puts another cloud into slot 0 makes the box stage's products stale (round-4 advisor): MOT_E_STATE, not cubes of the wrong cloud
        c.cluster(g["elevated"][: len(g["elevated"]) // 2])
        assert L.mot_box_markers(h, 0, six, 10, C.byref(n)) == mot.MOT_E_STATE and b"box stage" in L.mot_last_error(h)
        c.box_fit_resident()
        assert L.mot_box_markers(h, 0, None, 1024, C.byref(n)) == mot.MOT_OK
        c.ground_remove(cloud)
        assert L.mot_box_markers(h, 0, None, 1024, C.byref(n)) == mot.MOT_E_STATE
        sz = C.c_size_t(0); w = C.c_size_t(0); buf = (C.c_char * 16)()
        assert L.mot_stream_snapshot_size(h, None) == mot.MOT_E_ARG
        assert L.mot_stream_snapshot_size(h, C.byref(sz)) == mot.MOT_OK and sz.value > 32 * 1600
        assert L.mot_stream_save(h, 0, None, C.c_size_t(0), C.byref(w)) == mot.MOT_E_ARG
        assert L.mot_stream_save(h, 5, buf, C.c_size_t(16), C.byref(w)) == mot.MOT_E_ARG
        assert L.mot_stream_save(h, 0, buf, C.c_size_t(16), C.byref(w)) == mot.MOT_E_CAPACITY and w.value > 16
        assert L.mot_stream_load(h, 0, None, C.c_size_t(0)) == mot.MOT_E_ARG
        assert L.mot_stream_load(h, 0, buf, C.c_size_t(16)) == mot.MOT_E_ARG          # shorter than a header
        assert L.mot_set_tracker_mode(h, 7) == mot.MOT_E_ARG and b"mode" in L.mot_last_error(h)
        assert L.mot_set_tracker_mode(h, mot.MOT_TRACKER_SPLIT) == mot.MOT_OK and L.mot_set_tracker_mode(h, mot.MOT_TRACKER_AUTO) == mot.MOT_OK
        one = np.array([100], np.int32); ts = np.array([1.0e9]); z = np.zeros(1)
        pts = np.zeros((8192, 4), np.float32)
        seq = L.mot_sequence_dev
        assert seq(h, None, 8192 * 4, one.ctypes.data_as(C.c_void_p), 1, ts.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p), None, 0, None) == mot.MOT_E_ARG
        assert seq(h, pts.ctypes.data_as(C.c_void_p), 8192 * 4, one.ctypes.data_as(C.c_void_p), 1, None, z.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p), None, 0, None) == mot.MOT_E_ARG
        cnt = (C.c_int * 1)()
        assert seq(h, pts.ctypes.data_as(C.c_void_p), 8192 * 4, one.ctypes.data_as(C.c_void_p), 1, ts.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p), None, 4, cnt) == mot.MOT_E_ARG   # counts without records
        # ... and the context still works
        assert len(c.ground_remove(cloud)["elevated"]) == len(g["elevated"])


@pytest.mark.timeout(120)
def test_an_astronomical_timestamp_jump_does_not_hang_the_tracker(mot, emu):
    """dt = 1e34 s (a garbage timestamp) makes yaw + yaw_rate * dt an angle no subtraction of 2 pi can move: the reference's
    `while (a > M_PI) a -= 2. * M_PI` spins for ever there. The device answers NaN instead (track.hip: wrap_pi, as it does for Inf) and the step
    returns — with tracks killed by the divergence guards or carrying NaN, but it returns, on the emulator as on a GPU that must never hang"""
    import snapshot_case as S
    lib, L = emu
    with mot.Context(lib_path=lib, max_points=1024, max_batch=1, max_tracks_total=64) as c:
        for f in range(12):
            S._step(c, 0, f)
        assert int((c.get_tracks(0)["track_manage"] > 0).sum()) >= 4
        for jump in (1.0e40, 1.0e300, float("inf")):
            ts = 1.0e9 + 12 * 1e5 + jump
            c.ego_update(ts, 2.0, 0.1, 0)
            out = c.track_step(S.boxes_of(12), ts, 0)
            assert out["n"] >= 1
        c.reset_slot(0)
        assert S._step(c, 0, 0)["n"] == 1      # and the stream starts over cleanly


@pytest.mark.timeout(300)
def test_hostile_tracker_inputs_never_hang_or_crash(mot, emu):
    """boxes with NaN / Inf / 1e30 corners, zero-size and repeated boxes, timestamps that stand still, run backwards, jump or are NaN, ego values
    that are NaN or huge: the step must come back every time (no assertion on the numbers — the reference asserts or spins on most of these),
    the track count stays inside the context's limits and a reset gives a clean stream again"""
    import snapshot_case as S
    lib, L = emu
    rng = np.random.default_rng(77)
    specials = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 1e-30, 0.0, -0.0, 3.0e4, -3.0e4], np.float32)
    with mot.Context(lib_path=lib, max_points=1024, max_batch=1, max_tracks_total=32) as c:
        for trial in range(40):
            c.reset_slot(0)
            ts = 1.0e9
            for f in range(10):
                b = S.boxes_of(f + trial)
                kind = int(rng.integers(0, 6))
                if kind == 0:      # a few corners replaced by special values
                    idx = rng.integers(0, b.size, size=int(rng.integers(1, 12)))
                    b.reshape(-1)[idx] = rng.choice(specials, size=len(idx))
                elif kind == 1:    # every box collapsed to a point / all boxes identical
                    b[:] = b[:, :1, :] if rng.random() < 0.5 else b[:1]
                elif kind == 2:    # far away and huge
                    b *= np.float32(10.0 ** float(rng.integers(3, 30)))
                elif kind == 3:    # no boxes at all
                    b = b[:0]
                step = [1e5, 0.0, -1e5, 1e12, np.nan, 1e5][int(rng.integers(0, 6))]
                ts = ts + step if np.isfinite(step) else float("nan")
                v = float(rng.choice([2.0, 0.0, np.nan, 1e20, -5.0])); yaw = float(rng.choice([0.01 * f, np.nan, 1e10, -400.0]))
                c.ego_update(ts, v, yaw, 0)
                try:
                    out = c.track_step(b, ts, 0)
                    assert 0 <= out["n"] <= 32 * 64
                except mot.MotError as e:
                    assert e.code in (mot.MOT_E_CAPACITY, mot.MOT_E_ARG, mot.MOT_E_STATE)
                if not np.isfinite(ts):
                    ts = 1.0e9 + 1e6 * (trial + 1)
        c.reset_slot(0)
        assert S._step(c, 0, 0)["n"] == 1


@pytest.mark.timeout(600)
def test_hostile_clouds_never_hang_or_crash_the_stateless_stages(mot, emu, synth):
    """clouds salted with NaN / Inf / 1e30 / FLT_MAX coordinates, collapsed onto one point, stretched by 1e19, with special heights only — through the
    stage-wise calls (ground, cluster, box fit, cube markers, side products) and through a fused batch with the tracker: every call comes back
    with a status (run under MOT_EMU_SANITIZE=address this is the memory-safety check of every float -> cell / pixel conversion)"""
    lib, L = emu
    rng = np.random.default_rng(5)
    specials = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 1e-38, 0.0, -0.0, 3.4e38, -3.4e38, 59.99, 60.0, -60.0, 1e9], np.float32)
    with mot.Context(lib_path=lib, max_points=8192, max_batch=2, max_tracks_total=32) as c:
        for trial in range(36):
            n = int(rng.integers(0, 8000))
            base = synth.make_cloud(8000, int(rng.integers(0, 9)), trial)[:n].copy()
            kind = trial % 6
            if n and kind == 0:
                idx = rng.integers(0, base.size, size=int(rng.integers(1, 200))); base.reshape(-1)[idx] = rng.choice(specials, size=len(idx))
            elif n and kind == 1:
                base[:, :3] = base[0, :3]
            elif n and kind == 2:
                base[:, :2] *= np.float32(10.0 ** float(rng.integers(1, 20)))
            elif n and kind == 3:
                base[:, 2] = rng.choice(specials, size=n)
            elif n and kind == 4:
                base[:, :3] = rng.choice(specials, size=(n, 3))
            try:
                g = c.ground_remove(base); c.cluster(g["elevated"]); b = c.box_fit_resident()
                assert len(c.box_markers(0)) == len(b["boxes"])
                c.cluster_products(0)
                host = np.zeros((2, 8192, 4), np.float32); host[0, :n] = base; host[1, : n // 2] = base[: n // 2]
                c.frames_dev(host.ctypes.data, 8192 * 4, [n, n // 2], run_tracker=True, timestamps=[1e9 + trial * 1e5] * 2, ego_v=[1.0] * 2, ego_yaw=[0.0] * 2)
                assert len(c.box_markers(1)) == len(c.get_boxes(1)["boxes"])
                c.get_tracks(0)
            except mot.MotError as e:
                assert e.code in (mot.MOT_E_CAPACITY, mot.MOT_E_ARG, mot.MOT_E_STATE)


@pytest.mark.timeout(600)
def test_random_call_orders_never_crash(mot, emu, synth):
    """a few hundred valid calls in random order on one context — stage-wise stages, fused batches with and without the tracker, sequence mode,
    getters of both slots, resets, snapshots saved and loaded, output flags and tracker modes switched in between: every call returns a status
    (MOT_E_STATE where the header says so, e.g. mot_get_ground after a stage-wise call took slot 0), nothing crashes, the context stays usable"""
    import snapshot_case as S
    lib, L = emu
    rng = np.random.default_rng(11)
    N, stride = 6000, 6144
    clouds = [synth.make_cloud(N, s, f) for s, f in ((1, 0), (2, 1), (3, 2))]
    host = np.zeros((2, stride, 4), np.float32); host[0, :N] = clouds[0]; host[1, :N] = clouds[1]
    errs = {}
    with mot.Context(lib_path=lib, max_points=stride, max_batch=2, max_tracks_total=32) as c:
        elev = cl = blob = None; t = 0
        ops = ["ground", "cluster", "box_res", "box_fit", "markers", "products", "frames", "frames_trk", "get_ground", "get_clusters", "get_boxes", "get_tracks",
               "track", "reset", "reset_slot", "save", "load", "outputs", "mode", "seq", "products_host"]
        for _ in range(260):
            op = ops[int(rng.integers(0, len(ops)))]; slot = int(rng.integers(0, 2))
            try:
                if op == "ground": elev = c.ground_remove(clouds[int(rng.integers(0, 3))])["elevated"]
                elif op == "cluster" and elev is not None: cl = c.cluster(elev)
                elif op == "box_res": c.box_fit_resident()
                elif op == "box_fit" and cl is not None: c.box_fit(elev, cl["grid"], cl["num_cluster"])
                elif op == "markers": c.box_markers(slot)
                elif op == "products": c.cluster_products(slot)
                elif op == "products_host" and cl is not None: c.cluster_products_host(elev[:3000], cl["grid"])
                elif op == "frames": c.frames_dev(host.ctypes.data, stride * 4, [N, N - int(rng.integers(0, N))])
                elif op == "frames_trk":
                    t += 1; c.frames_dev(host.ctypes.data, stride * 4, [N, N], run_tracker=True, timestamps=[1e9 + t * 1e5] * 2, ego_v=[1.0] * 2, ego_yaw=[0.0] * 2)
                elif op == "get_ground": c.get_ground(slot, n_hint=N)
                elif op == "get_clusters": c.get_clusters(slot, int(rng.integers(0, N)))
                elif op == "get_boxes": c.get_boxes(slot)
                elif op == "get_tracks": c.get_tracks(slot)
                elif op == "track":
                    t += 1; c.ego_update(1e9 + t * 1e5, 1.0, 0.0, slot); c.track_step(S.boxes_of(t % 30), 1e9 + t * 1e5, slot)
                elif op == "reset": c.reset()
                elif op == "reset_slot": c.reset_slot(slot) if rng.random() < 0.5 else c.reset_tracks_slot(slot)
                elif op == "save": blob = c.stream_save(slot)
                elif op == "load" and blob is not None: c.stream_load(slot, blob)
                elif op == "outputs": c.set_fused_outputs(int(rng.integers(0, 8)))
                elif op == "mode": c.set_tracker_mode(int(rng.integers(0, 3)))
                elif op == "seq":
                    t += 2
                    c.sequence_dev(host.ctypes.data, stride * 4, np.array([N, N], np.int32), np.array([1e9 + (t - 1) * 1e5, 1e9 + t * 1e5]), np.array([1.0, 1.0]), np.array([0.0, 0.0]))
            except mot.MotError as e:
                errs[(op, e.code)] = errs.get((op, e.code), 0) + 1
        c.synchronize()
        assert all(code in (mot.MOT_E_STATE, mot.MOT_E_CAPACITY) for _, code in errs), errs
        c.reset(); c.set_fused_outputs(0); c.set_tracker_mode(0)
        assert len(c.ground_remove(clouds[0])["elevated"]) > 100


@pytest.mark.timeout(600)
def test_extreme_parameters_are_refused_or_harmless(mot, emu, synth):
    """mot_params with one to three fields set to 0, -1, NaN, Inf, 1e30, INT_MAX ...: mot_create either refuses them or the context runs a frame
    and a few tracker steps to the end (a parameter must never size a buffer the kernels then overrun, nor a loop that does not end;
    MOT_EMU_SANITIZE=address makes this the memory-safety check of the parameter validation)"""
    import snapshot_case as S
    lib, L = emu
    rng = np.random.default_rng(3)
    cloud = synth.make_cloud(6000, 2, 0)
    fvals = [0.0, -1.0, 1e-30, 1e30, float("nan"), float("inf"), -float("inf"), 1.0, 0.5, 100.0, -100.0, 1e9]
    ivals = [0, -1, 1, 2, 3, 7, 64, 255, 256, 257, 1000, 65536, 2 ** 31 - 1, -2 ** 31]
    types = dict(mot.MotParams._fields_); fields = list(types)
    created = refused = 0
    for trial in range(90):
        p = mot.params(trial % 2, lib=L)
        for _ in range(int(rng.integers(1, 4))):
            f = fields[int(rng.integers(0, len(fields)))]
            setattr(p, f, ivals[int(rng.integers(0, len(ivals)))] if types[f] is C.c_int32 else fvals[int(rng.integers(0, len(fvals)))])
        try:
            c = mot.Context(p, lib_path=lib, max_points=6144, max_batch=1, max_tracks_total=16)
        except mot.MotError as e:
            assert e.code == mot.MOT_E_ARG; refused += 1
            continue
        created += 1
        try:
            g = c.ground_remove(cloud); c.cluster(g["elevated"]); c.box_fit_resident(); c.box_markers(0); c.cluster_products(0)
            for f in range(4):
                S._step(c, 0, f)
        except mot.MotError as e:
            assert e.code in (mot.MOT_E_CAPACITY, mot.MOT_E_ARG, mot.MOT_E_STATE)
        c.close()
    assert created >= 30 and refused >= 5, (created, refused)
