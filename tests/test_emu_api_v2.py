"""Host logic of the ABI-v2 entry points, exercised on the emulator build (tests/emu/hipemu.h: the same csrc/*.hip compiled by
g++; development check of LOGIC, not a parity claim — the -m gpu tests repeat the comparisons on the MI355X):
pipelined host ingest, batched device-box tracker step, in-run kernel timing, per-slot reset, sticky track capacity,
capacity-checked getters."""
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


def _moving_boxes(f, m=6):
    b = np.zeros((m, 8, 3), np.float32)
    for k in range(m):
        b[k, :, :2] = np.array([[0, 0], [2, 0], [2, 1], [0, 1]] * 2) + [6.0 * k - 15 + 0.3 * f, 4.0 + 0.1 * f * (k % 3)]
        b[k, :4, 2] = -2.0; b[k, 4:, 2] = 0.5
    return b


def test_frames_host_equals_frames_dev(mot, emu, synth, oracle):
    lib, L = emu
    B, N, stride = 3, 5000, 5120
    p = oracle.params(0)
    with mot.Context(lib_path=lib, max_points=stride, max_batch=B, max_tracks_total=128) as a, \
         mot.Context(lib_path=lib, max_points=stride, max_batch=B, max_tracks_total=128) as b:
        for f in range(5):   # more calls than staging buffers: both are reused
            n = [N, N - 777, 64 + f]
            host = np.zeros((B, stride, 4), np.float32)
            for s in range(B):
                host[s, : n[s]] = synth.make_cloud(N, 40 + s, f)[: n[s]]
            ts = [1.0e9 + f * 1e5] * B
            kw = dict(run_tracker=True, timestamps=ts, ego_v=[1.0] * B, ego_yaw=[0.01 * f] * B)
            a.frames_dev(host.ctypes.data, stride * 4, n, **kw)
            # once dense (frames at the staging stride: one copy), once ragged (per-frame copies)
            if f % 2 == 0:
                b.frames_host(host.ctypes.data, stride * 4, n, **kw)
            else:
                wide = np.zeros((B, stride + 64, 4), np.float32); wide[:, :stride] = host
                b.frames_host(wide.ctypes.data, (stride + 64) * 4, n, **kw)
            b.wait_uploads()
            for s in range(B):
                ga, gb = a.get_ground(s, n_hint=n[s]), b.get_ground(s, n_hint=n[s])
                assert np.array_equal(ga["elevated"], gb["elevated"]) and np.array_equal(ga["mask"], gb["mask"])
                assert np.array_equal(a.get_boxes(s)["boxes"], b.get_boxes(s)["boxes"])
                ta, tb = a.get_tracks(s), b.get_tracks(s)
                assert ta["n"] == tb["n"] and np.array_equal(ta["track_manage"], tb["track_manage"]) and np.array_equal(ta["p"], tb["p"])
            # and against the oracle for one stream
            g = oracle.ground_remove(p, host[0, : n[0]])
            assert np.array_equal(b.get_ground(0, n_hint=n[0])["elevated"], g["elevated"])
        # the fetched host block = the live tracks of get_tracks
        K = 16
        rec = np.zeros((B, K, 36), np.int32); cnt = np.zeros(B, np.int32)
        b.fetch_tracks_async(B, rec.ctypes.data, K, cnt.ctypes.data); b.synchronize()
        for s in range(B):
            t = b.get_tracks(s)
            live = np.nonzero(t["track_manage"] > 0)[0][:K]
            assert cnt[s] == len(live) and np.array_equal(rec[s, : cnt[s], 0], live)


def test_frames_host_xyz_equals_frames_host(mot, emu, synth, oracle):
    """mot_frames_host_xyz (ABI v6): packed {x, y, z} records from host memory, expanded on the device with w = 1.0f — every result equals the float4
    entry point's on the same x, y, z (dense and ragged strides, more calls than staging buffers), and the oracle's"""
    lib, L = emu
    B, N, stride = 3, 5000, 5120
    p = oracle.params(0)
    with mot.Context(lib_path=lib, max_points=stride, max_batch=B, max_tracks_total=128) as a, \
         mot.Context(lib_path=lib, max_points=stride, max_batch=B, max_tracks_total=128) as b:
        for f in range(5):
            n = [N, N - 777, 64 + f]
            host = np.zeros((B, stride, 4), np.float32)
            for s in range(B):
                host[s, : n[s]] = synth.make_cloud(N, 50 + s, f)[: n[s]]
            host[..., 3] = 1.0
            kw = dict(run_tracker=True, timestamps=[1.0e9 + f * 1e5] * B, ego_v=[1.0] * B, ego_yaw=[0.01 * f] * B)
            a.frames_host(host.ctypes.data, stride * 4, n, **kw)
            if f % 2 == 0:
                xyz = np.ascontiguousarray(host[..., :3])
                b.frames_host_xyz(xyz.ctypes.data, stride * 3, n, **kw)
            else:
                xyz = np.zeros((B, stride + 10, 3), np.float32); xyz[:, :stride] = host[..., :3]
                b.frames_host_xyz(xyz.ctypes.data, (stride + 10) * 3, n, **kw)
            a.wait_uploads(); b.wait_uploads()
            for s in range(B):
                ga, gb = a.get_ground(s, n_hint=n[s]), b.get_ground(s, n_hint=n[s])
                assert np.array_equal(ga["elevated"].view(np.uint32), gb["elevated"].view(np.uint32)) and np.array_equal(ga["ground"].view(np.uint32), gb["ground"].view(np.uint32))
                assert np.array_equal(ga["mask"], gb["mask"]) and (gb["elevated"][:, 3] == 1.0).all()
                assert np.array_equal(a.get_boxes(s)["boxes"], b.get_boxes(s)["boxes"]) and np.array_equal(a.get_clusters(s)["grid"], b.get_clusters(s)["grid"])
                ta, tb = a.get_tracks(s), b.get_tracks(s)
                assert ta["n"] == tb["n"] and np.array_equal(ta["track_manage"], tb["track_manage"]) and np.array_equal(ta["p"], tb["p"])
            g = oracle.ground_remove(p, host[0, : n[0]])
            assert np.array_equal(b.get_ground(0, n_hint=n[0])["elevated"], g["elevated"])
        # argument errors: before anything is queued
        with pytest.raises(mot.MotError) as e:
            b.frames_host_xyz(xyz.ctypes.data, 30, [100, 100])   # stride smaller than a frame
        assert e.value.code == mot.MOT_E_ARG
        with pytest.raises(mot.MotError) as e:
            b.frames_host_xyz(xyz.ctypes.data, stride * 3, [stride + 1])
        assert e.value.code == mot.MOT_E_CAPACITY


def pointcloud2_payload(cloud, step, ox, oy, oz, ow, seed):
    """a sensor_msgs/PointCloud2 `data` block: records of `step` bytes with float32 fields at the given offsets, everything else random bytes"""
    n = len(cloud)
    raw = np.random.default_rng(seed).integers(0, 256, size=(n, step), dtype=np.uint8)
    for off, col in ((ox, 0), (oy, 1), (oz, 2)) + (((ow, 3),) if ow >= 0 else ()):
        raw[:, off:off + 4] = np.ascontiguousarray(cloud[:, col], np.float32).view(np.uint8).reshape(n, 4)
    return np.ascontiguousarray(raw.reshape(-1))


@pytest.mark.parametrize("step,ox,oy,oz,ow", [(16, 0, 4, 8, 12), (32, 0, 4, 8, 16), (22, 0, 4, 8, 12), (22, 10, 2, 6, -1), (12, 0, 4, 8, -1)])
def test_frames_host_pointcloud2_equals_frames_host(mot, emu, synth, oracle, step, ox, oy, oz, ow):
    """mot_frames_host_pointcloud2 (ABI v6): one PointCloud2 payload per stream, each in its own host buffer, any point_step (16: kitti2bag; 32 / 22: a velodyne
    driver's records, the second one with unaligned fields), unpacked on the device — every result equals mot_frames_host's on the decoded clouds"""
    lib, L = emu
    B, N, stride = 3, 4000, 4096
    with mot.Context(lib_path=lib, max_points=stride, max_batch=B, max_tracks_total=128) as a, \
         mot.Context(lib_path=lib, max_points=stride, max_batch=B, max_tracks_total=128) as b:
        for f in range(4):   # more calls than staging buffers
            n = [N, N - 555, 0 if f == 2 else 40 + f]
            host = np.zeros((B, stride, 4), np.float32)
            for s in range(B):
                host[s, : n[s]] = synth.make_cloud(N, 60 + s, f)[: n[s]]
            if ow < 0:
                host[..., 3] = 1.0
            kw = dict(run_tracker=True, timestamps=[1.0e9 + f * 1e5] * B, ego_v=[1.0] * B, ego_yaw=[0.01 * f] * B)
            a.frames_host(host.ctypes.data, stride * 4, n, **kw)
            payloads = [pointcloud2_payload(host[s, : n[s]], step, ox, oy, oz, ow, 7 * f + s) if n[s] else None for s in range(B)]
            b.frames_host_pointcloud2(payloads, n, step, ox, oy, oz, ow, **kw)
            a.wait_uploads(); b.wait_uploads()
            for s in range(B):
                ga, gb = a.get_ground(s, n_hint=max(n[s], 1)), b.get_ground(s, n_hint=max(n[s], 1))
                assert np.array_equal(ga["elevated"].view(np.uint32), gb["elevated"].view(np.uint32)) and np.array_equal(ga["ground"].view(np.uint32), gb["ground"].view(np.uint32))
                assert np.array_equal(ga["mask"], gb["mask"]) and np.array_equal(a.get_boxes(s)["boxes"], b.get_boxes(s)["boxes"])
                ta, tb = a.get_tracks(s), b.get_tracks(s)
                assert ta["n"] == tb["n"] and np.array_equal(ta["track_manage"], tb["track_manage"]) and np.array_equal(ta["p"], tb["p"])
        with pytest.raises(mot.MotError) as e:
            b.frames_host_pointcloud2(payloads, n, step, ox, step - 2, oz, ow)   # a field that sticks out of the record
        assert e.value.code == mot.MOT_E_ARG
        with pytest.raises(mot.MotError) as e:
            b.frames_host_pointcloud2(payloads, [stride + 1, 1, 1], step, ox, oy, oz, ow)
        assert e.value.code == mot.MOT_E_CAPACITY


def test_track_steps_dev_equals_track_step(mot, emu):
    lib, L = emu
    B = 3
    with mot.Context(lib_path=lib, max_points=1024, max_batch=B, max_tracks_total=64) as a, \
         mot.Context(lib_path=lib, max_points=1024, max_batch=B, max_tracks_total=64) as b:
        for f in range(8):
            ts = 1.0e9 + f * 1e5
            m = [6, 3 + f % 3, 0 if f == 4 else 5]
            stride = 8 * 24
            blk = np.zeros((B, stride), np.float32)
            for s in range(B):
                bx = _moving_boxes(f + s)[: m[s]]
                blk[s, : m[s] * 24] = bx.reshape(-1)
                a.ego_update(ts, 0.5, 0.0, s); b.ego_update(ts, 0.5, 0.0, s)
                a.track_step(bx, ts, s)
            b.track_steps_dev(blk.ctypes.data, stride, m, [ts] * B)
            for s in range(B):
                ta, tb = a.get_tracks(s), b.get_tracks(s)
                assert ta["n"] == tb["n"] and np.array_equal(ta["track_manage"], tb["track_manage"])
                assert np.array_equal(ta["p"], tb["p"]) and np.array_equal(ta["v_yaw"], tb["v_yaw"])
        with pytest.raises(mot.MotError) as e:
            b.track_steps_dev(blk.ctypes.data, 24, [2, 0, 0], [ts + 1e5] * B)   # two boxes do not fit a 24-float stride
        assert e.value.code == mot.MOT_E_ARG


def test_reset_slot_and_sticky_capacity(mot, emu):
    lib, L = emu
    with mot.Context(lib_path=lib, max_points=1024, max_batch=2, max_tracks_total=4) as c:
        hit = 0
        for f in range(6):
            ts = 1.0e9 + f * 1e5
            for s in range(2):
                c.ego_update(ts, 0.0, 0.0, s)
                out = c.track_step(_moving_boxes(f, 12), ts, s)   # the records are delivered; the binding reports the condition softly
                hit += int(out["capacity_exceeded"])
                assert int((out["track_manage"] > 0).sum()) <= 4   # 4 track SLOTS: at most 4 alive (tracks ever created may be more: a dead one frees its slot a step later)
        assert hit >= 6                      # told again on every call once births are being dropped (both streams)
        tr = (mot.MotTrack * 64)(); nt = C.c_int(0)
        assert L.mot_get_tracks(c._h, 0, tr, 64, C.byref(nt)) == mot.MOT_E_CAPACITY and 4 <= nt.value <= 64   # the C call: MOT_E_CAPACITY, records delivered
        assert sum(1 for i in range(nt.value) if tr[i].track_manage > 0) <= 4 and all(tr[i].id == i for i in range(nt.value))
        assert c.get_tracks(0)["capacity_exceeded"]   # ... also by the getter, until the stream is started over
        c.reset_slot(0)
        t0 = c.get_tracks(0)
        assert t0["n"] == 0 and not t0["capacity_exceeded"]     # slot 0 forgot everything
        assert c.get_tracks(1)["capacity_exceeded"]             # slot 1 did not
        with pytest.raises(mot.MotError) as e:
            c.reset_slot(5)
        assert e.value.code == mot.MOT_E_ARG and "slot" in str(e.value)


def test_getters_check_caller_capacity(mot, emu, synth):
    lib, L = emu
    with mot.Context(lib_path=lib, max_points=4096, max_batch=1, max_tracks_total=8) as c:
        cloud = synth.make_cloud(4000, 2, 0)
        c.frames_dev(cloud.ctypes.data, 4096 * 4, [4000])
        ne = c.get_ground(0, want_clouds=False)["n_elevated"]
        assert ne > 10
        small = np.zeros((ne - 1, 4), np.float32); n_e = C.c_int(0)
        rc = L.mot_get_ground(c._h, 0, small.ctypes.data_as(C.c_void_p), C.byref(n_e), None, None, None, ne - 1)
        assert rc == mot.MOT_E_CAPACITY and n_e.value == ne and b"capacity_points" in L.mot_last_error(c._h)
        lab = np.zeros(ne - 1, np.int32)
        rc = L.mot_get_clusters(c._h, 0, None, None, lab.ctypes.data_as(C.c_void_p), ne - 1)
        assert rc == mot.MOT_E_CAPACITY
        lab = np.zeros(ne, np.int32)
        assert L.mot_get_clusters(c._h, 0, None, None, lab.ctypes.data_as(C.c_void_p), ne) == mot.MOT_OK
        # a cloud that is not 16-byte aligned is refused before anything is launched
        raw = np.zeros(4096 * 4 + 1, np.float32)
        off = raw[1:] if raw.ctypes.data % 16 == 0 else raw[:-1]
        if off.ctypes.data % 16:
            rc = L.mot_frames_dev(c._h, C.c_void_p(off.ctypes.data), C.c_long(4096 * 4), np.array([10], np.int32).ctypes.data_as(C.c_void_p), 1, 0, None, None, None)
            assert rc == mot.MOT_E_ARG and b"aligned" in L.mot_last_error(c._h)


def test_profile_ring(mot, emu, synth):
    lib, L = emu
    with mot.Context(lib_path=lib, max_points=2048, max_batch=1, max_tracks_total=8) as c:
        cloud = synth.make_cloud(2000, 2, 0)
        c.profile_kernel(12)                 # the compaction kernel
        for _ in range(3):
            c.frames_dev(cloud.ctypes.data, 2048 * 4, [2000])
        r = c.profile_read()
        assert r["samples"] == 3 and r["mean_ms"] >= 0 and r["min_ms"] <= r["mean_ms"] <= r["max_ms"]
        assert c.profile_read()["samples"] == 0
        c.profile_kernel(0)
        c.frames_dev(cloud.ctypes.data, 2048 * 4, [2000])
        assert c.profile_read()["samples"] == 0


def test_fast_path_sweeps_on_the_emulator(mot, emu):
    """the sweep hook itself (the real check needs the hardware's v_sqrt / v_rcp: tests/test_ground_gpu.py)"""
    lib, L = emu
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "devcheck"))
    import build_sweep
    S = C.CDLL(build_sweep.build_emu())
    with mot.Context(lib_path=lib, max_points=1024) as c:
        dp = (C.c_char * 512)()
        assert L.mot_debug_dev_params(c._h, dp, C.c_size_t(512)) == 0
        st = (C.c_ulonglong * 8)()
        for what in (0, 1, 2):
            for mode in (0, 1, 2):
                assert S.mot_sweep_run(dp, what, mode, C.c_ulonglong(7), C.c_ulonglong(200000), st) == 0
                assert st[0] == 200000 and st[2] == 0, (what, mode, list(st))


def test_fused_outputs_on_demand_and_materialised(mot, emu, synth, oracle):
    """mot_set_fused_outputs: by default the fused path writes neither the ground cloud nor the mask nor the per-point cluster labels
    (nothing downstream reads them); mot_get_ground re-runs the compaction of the last batch when they are asked for, mot_get_clusters
    computes a slot's labels from its cells and label grid. With the flags set they are written in the first place. Either way: equal to the oracle, for every slot of a ragged batch, and the cluster stage's results are untouched."""
    lib, L = emu
    B, N, stride = 3, 5000, 5120
    p = oracle.params(0)
    n = [N, N - 1234, 300]
    host = np.zeros((B, stride, 4), np.float32)
    for s in range(B):
        host[s, : n[s]] = synth.make_cloud(N, 50 + s, 1)[: n[s]]
    res = {}
    for flags in (0, mot.OUT_GROUND, mot.OUT_GROUND | mot.OUT_MASK, mot.OUT_LABELS, mot.OUT_GROUND | mot.OUT_MASK | mot.OUT_LABELS):
        with mot.Context(lib_path=lib, max_points=stride, max_batch=B, max_tracks_total=64) as c:
            c.set_fused_outputs(flags)
            c.frames_dev(host.ctypes.data, stride * 4, n)
            bx = [c.get_boxes(s)["boxes"] for s in range(B)]          # before any materialisation
            cl = [c.get_clusters(s, n_elevated=n[s]) for s in range(B)]
            g = [c.get_ground(s, n_hint=n[s]) for s in (2, 0, 1)]      # any order; one re-run serves the whole batch
            g = [g[1], g[2], g[0]]
            assert all(np.array_equal(c.get_boxes(s)["boxes"], bx[s]) for s in range(B))   # ... and leaves the later stages' results alone
            res[flags] = (g, bx, cl)
    for s in range(B):
        o = oracle.ground_remove(p, host[s, : n[s]])
        for flags, (g, bx, cl) in res.items():
            assert np.array_equal(g[s]["mask"], o["mask"]) and np.array_equal(g[s]["elevated"], o["elevated"]) and np.array_equal(g[s]["ground"], o["ground"]), (flags, s)
            assert np.array_equal(bx[s], res[0][1][s]) and np.array_equal(cl[s]["grid"], res[0][2][s]["grid"])
            assert np.array_equal(cl[s]["point_label"][: len(o["elevated"])], oracle.cluster(p, o["elevated"])["point_label"])
    with mot.Context(lib_path=lib, max_points=stride, max_batch=1) as c:
        with pytest.raises(mot.MotError) as e:
            c.set_fused_outputs(8)
        assert e.value.code == mot.MOT_E_ARG


def test_point_labels_on_demand_without_cells(mot, emu, synth, oracle):
    """the other two ways into the on-demand labels: a 256-cell grid (all 65536 cell codes in use, so the compaction kernel hands no
    cells over: the label comes from the point itself) and a stage-wise mot_cluster that was not asked for labels"""
    lib, L = emu
    N, stride = 6000, 6144
    cloud = synth.make_cloud(N, 77, 2)
    host = np.zeros((1, stride, 4), np.float32); host[0, :N] = cloud
    po = oracle.params(0, num_grid=256)
    o = oracle.ground_remove(po, cloud)
    want = oracle.cluster(po, o["elevated"])
    assert want["num_cluster"] > 3
    with mot.Context(mot.params(0, lib=mot.load_library(lib), num_grid=256), lib_path=lib, max_points=stride, max_batch=1) as c:
        c.frames_dev(host.ctypes.data, stride * 4, [N])
        got = c.get_clusters(0, n_elevated=len(o["elevated"]))
        assert np.array_equal(got["grid"], want["grid"]) and np.array_equal(got["point_label"], want["point_label"])
    po = oracle.params(0)
    want = oracle.cluster(po, o["elevated"])
    with mot.Context(lib_path=lib, max_points=stride, max_batch=1) as c:
        c.frames_dev(host.ctypes.data, stride * 4, [N])                    # slot 0 holds a fused result (cells resident) ...
        a = np.ascontiguousarray(o["elevated"][::-1])                       # ... then a stage-wise cloud in another order takes its place
        G = c.params.num_grid
        grid = np.zeros((G, G), np.int32); nc = mot.C.c_int(0)
        assert L.mot_cluster(c._h, a.ctypes.data_as(mot.C.c_void_p), len(a), grid.ctypes.data_as(mot.C.c_void_p), mot.C.byref(nc), None) == 0
        got = c.get_clusters(0, n_elevated=len(a))
        ref = oracle.cluster(po, a)
        assert np.array_equal(got["grid"], ref["grid"]) and np.array_equal(got["point_label"], ref["point_label"])
        assert ref["num_cluster"] == want["num_cluster"]


def test_stage_wise_call_after_a_fused_batch_leaves_the_other_slots_readable(mot, emu, synth, oracle):
    import mixed_use_case
    mixed_use_case.run(mot, emu[0], synth, oracle)
    mixed_use_case.run_ground_after_takeover(mot, emu[0], synth, oracle)


def test_reset_tracks_slot_keeps_the_global_frame(mot, emu, oracle):
    """mot_reset_tracks_slot forgets a stream's tracks but not its ego dead reckoning: the origin of the global frame stays where it
    was (mot_reset_slot would re-origin it at the current pose), and the next tracker step seeds anew like the reference's first frame"""
    lib, L = emu
    with mot.Context(lib_path=lib, max_points=1024, max_batch=2, max_tracks_total=64) as c, mot.Context(lib_path=lib, max_points=1024, max_batch=1, max_tracks_total=64) as ref:
        for f in range(6):
            ts = 1.0e9 + f * 1e5
            e0 = c.ego_update(ts, 3.0, 0.02 * f, 0); r0 = ref.ego_update(ts, 3.0, 0.02 * f)
            assert np.array_equal(e0, r0)
            c.track_step(_moving_boxes(f), ts, 0); ref.track_step(_moving_boxes(f), ts)
        assert c.get_tracks(0)["n"] > 1
        c.reset_tracks_slot(0)
        assert c.get_tracks(0)["n"] == 0
        for f in range(6, 10):
            ts = 1.0e9 + f * 1e5
            e0 = c.ego_update(ts, 3.0, 0.02 * f, 0); r0 = ref.ego_update(ts, 3.0, 0.02 * f)
            assert np.array_equal(e0, r0) and abs(e0[0]) + abs(e0[1]) > 0.5     # same pose as the uninterrupted stream: no re-origin
            t = c.track_step(_moving_boxes(f), ts, 0); ref.track_step(_moving_boxes(f), ts)
            if f == 6:
                assert t["n"] == 1 and t["track_manage"][0] == 1             # the reference's first frame: one seeded track
        assert c.get_tracks(0)["n"] > 1


def test_stream_snapshot_round_trip(mot, emu):
    """mot_stream_save / mot_stream_load: a stream moved to another slot of another context continues bit for bit (tests/snapshot_case.py)"""
    import snapshot_case
    snapshot_case.check(mot, emu[0])


def test_stream_load_survives_damaged_snapshots(mot, emu):
    """a snapshot with random bytes flipped — in its header, its index arrays, its filter states — is either refused or loaded; in both cases the
    context keeps stepping (whatever garbage the tracker then computes stays inside its buffers: run under MOT_EMU_SANITIZE=address this is a
    memory-safety check of mot_stream_load's validation and of the kernels on corrupt state)"""
    import snapshot_case as S
    lib = emu[0]
    rng = np.random.default_rng(2024)
    with mot.Context(lib_path=lib, max_points=1024, max_batch=2, max_tracks_total=64) as a, mot.Context(lib_path=lib, max_points=1024, max_batch=2, max_tracks_total=64) as b:
        for f in range(18):
            S._step(a, 0, f)
        blob = a.stream_save(0)
        hb, tb, T = (int(v) for v in np.frombuffer(blob[:24], np.uint32)[[2, 3, 5]])
        regions = [(0, hb), (hb, hb + 64 * 8), (hb + T * tb, hb + T * tb + 8 * T + 8), (len(blob) - 400, len(blob)), (0, len(blob))]
        refused = loaded = 0
        for k in range(120):
            lo, hi = regions[k % len(regions)]
            bad = bytearray(blob)
            for _ in range(1 + k % 4):
                i = int(rng.integers(lo, hi)); bad[i] = int(rng.integers(0, 256))
            try:
                b.stream_load(1, bytes(bad)); loaded += 1
            except mot.MotError as e:
                assert e.code in (mot.MOT_E_ARG, mot.MOT_E_CAPACITY); refused += 1
            for f in (18, 19):
                ts = 1.0e9 + f * 1e5
                b.ego_update(ts, 2.0, 0.01 * f, 1)
                try:
                    b.track_step(S.boxes_of(f), ts, 1)
                except mot.MotError:
                    pass
            b.reset_slot(1)
        assert refused >= 10 and loaded >= 10, (refused, loaded)
        # and an undamaged one still loads and continues like the original
        b.stream_load(1, blob)
        S._same(S._step(a, 0, 18), S._step(b, 1, 18))


def test_trace_ranges_are_optional(mot, emu_lib=None):
    """mot_set_trace_ranges: roctx ranges around the stages. The emulator build has no roctx: the call must answer MOT_E_STATE and leave
    everything working (the real library loads libroctx64 lazily; tests/test_api_v2_gpu.py turns the ranges on around a frame)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    import build_emu
    with mot.Context(lib_path=build_emu.build(), max_points=2048) as c:
        assert c.lib.mot_set_trace_ranges(c._h, 1) == mot.MOT_E_STATE and b"roctx" in c.lib.mot_last_error(c._h)
        assert c.lib.mot_set_trace_ranges(c._h, 0) == mot.MOT_OK
