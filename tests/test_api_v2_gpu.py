"""ABI-v2 entry points on the MI355X, through the C-ABI: the on-device proof of the guarded fast cells (real v_sqrt_f32 /
v_rcp_f32), pipelined host ingest == resident path == oracle, batched device-box tracker step, polar-grid intermediates."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import hiprt

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "devcheck"))

pytestmark = pytest.mark.gpu


def test_fast_cells_agree_with_exact_on_device(mot, hip_lib):
    """getCellIndexFromPoints (ground_removal.cpp:67-76) and mapCartesianGrid's index
    (component_clustering.cpp:42-48). The streaming kernels answer from estimates built on the hardware's 1-ulp v_sqrt_f32 /
    v_rcp_f32; whenever they answer (not -2) the answer must be the exact evaluation's. > 4e9 points: uniform random, a
    2^16 x 2^16 lattice, and every channel spoke / bin ring / grid line moved by -3..+3 steps of 1..64 ulp in x and y, both presets."""
    import build_sweep   # tests/devcheck: the sweep kernels are a test library of their own (until round 4 they shipped inside libmot_hip.so)
    S = C.CDLL(build_sweep.build())
    total = 0
    for preset in (0, 1):
        with mot.Context(mot.params(preset), max_points=1024) as c:
            dp = (C.c_char * 512)()
            assert hip_lib.mot_debug_dev_params(c._h, dp, C.c_size_t(512)) == 0
            st = (C.c_ulonglong * 8)()
            for what, mode, count in ((0, 0, 1 << 30), (0, 1, 1 << 30), (0, 2, 1 << 29), (1, 0, 1 << 28), (1, 1, 1 << 28), (1, 2, 1 << 28), (2, 0, 1 << 29), (2, 1, 1 << 29), (2, 2, 1 << 29)):
                if preset == 1 and what in (0, 2) and mode < 2:
                    continue   # the polar grid does not depend on the preset: boundaries only
                rc = S.mot_sweep_run(dp, what, mode, C.c_ulonglong(1234567 + 17 * mode + preset), C.c_ulonglong(count), st)
                assert rc == 0
                assert st[0] == count
                x = np.array([st[3] & 0xffffffff], np.uint32).view(np.float32)[0]; y = np.array([st[3] >> 32], np.uint32).view(np.float32)[0]
                assert st[2] == 0, f"what={what} mode={mode} preset={preset}: {st[2]} mismatches, first at ({x!r}, {y!r}) fast/exact {st[4] & 0xffffffff:#x}/{st[4] >> 32:#x}"
                und = st[1] / count
                assert und < (0.999 if mode == 2 else 3e-3), (what, mode, und)   # the exact path stays rare away from boundaries (mode 2 sits ON them)
                total += count
    assert total > 4e9


def test_polar_grid_intermediates_on_device(mot, hip_lib, oracle, synth):
    """a4-a8 (clamp, Gaussian, hDiff, decision, median, outlier): the per-cell ground thresholds the filter kernel leaves in
    HBM equal the oracle's hGround on ground cells and are -inf elsewhere"""
    p = oracle.params(0)
    with mot.Context(max_points=131072) as c:
        for stream in (0, 3, 9):
            cloud = synth.make_cloud(120000, stream, 1)
            c.ground_remove(cloud)
            hg = np.zeros(80 * 120, np.float32)
            assert hip_lib.mot_debug_copy(c._h, 10, 0, hg.ctypes.data_as(C.c_void_p), C.c_size_t(hg.nbytes)) == 0
            d = oracle.ground_remove(p, cloud, want_dump=True)
            isg = d["is_ground"].reshape(-1).astype(bool)
            assert np.array_equal(np.isfinite(hg), isg)
            assert np.array_equal(hg[isg].view(np.uint32), d["hground"].astype(np.float32).reshape(-1)[isg].view(np.uint32))


def test_frames_host_equals_resident_path_and_oracle(mot, hip_lib, oracle, synth):
    B, N, stride = 4, 60000, 61440
    p = oracle.params(0)
    hp = C.c_void_p()
    assert hip_lib.mot_host_alloc(C.c_size_t(B * stride * 16), C.byref(hp)) == 0
    pinned = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_float)), shape=(B, stride, 4))
    with mot.Context(max_points=stride, max_batch=B, max_tracks_total=256) as a, mot.Context(max_points=stride, max_batch=B, max_tracks_total=256) as b:
        T = oracle.Tracker(p)
        for f in range(6):
            n = [N, N - 1234, 2048, 7]
            host = np.zeros((B, stride, 4), np.float32)
            for s in range(B):
                host[s, : n[s]] = synth.make_cloud(N, 70 + s, f)[: n[s]]
            ts = [1.0e9 + f * 1e5] * B
            kw = dict(run_tracker=True, timestamps=ts, ego_v=[2.0] * B, ego_yaw=[0.002 * f] * B)
            dev = hiprt.DeviceBuffer(host)
            a.frames_dev(dev.ptr, stride * 4, n, **kw)
            b.wait_uploads()          # the pinned block is refilled: the previous upload must have left it
            pinned[:] = host
            b.frames_host(hp.value, stride * 4, n, **kw)
            for s in range(B):
                ga, gb = a.get_ground(s, n_hint=n[s]), b.get_ground(s, n_hint=n[s])
                assert np.array_equal(ga["elevated"].view(np.uint32), gb["elevated"].view(np.uint32)) and np.array_equal(ga["mask"], gb["mask"])
                assert np.array_equal(a.get_boxes(s)["boxes"], b.get_boxes(s)["boxes"])
                ta, tb = a.get_tracks(s), b.get_tracks(s)
                assert ta["n"] == tb["n"] and np.array_equal(ta["track_manage"], tb["track_manage"]) and np.array_equal(ta["p"], tb["p"])
            g = oracle.ground_remove(p, host[0, : n[0]]); cl = oracle.cluster(p, g["elevated"])
            bx = oracle.box_fit(p, g["elevated"], cl["grid"], cl["num_cluster"])["boxes"]
            assert np.array_equal(b.get_ground(0, n_hint=n[0])["elevated"].view(np.uint32), g["elevated"].view(np.uint32))
            assert np.array_equal(b.get_clusters(0)["grid"], cl["grid"]) and np.array_equal(b.get_boxes(0)["boxes"], bx)
            a.synchronize()
    T.close()
    assert hip_lib.mot_host_free(hp) == 0


def test_frames_host_xyz_equals_frames_host_and_oracle(mot, hip_lib, oracle, synth):
    """mot_frames_host_xyz on the MI355X: packed 12-byte records from page-locked memory against the float4 entry point on the same x, y, z, and the oracle"""
    B, N, stride = 4, 60000, 61440
    p = oracle.params(0)
    hp, hq = C.c_void_p(), C.c_void_p()
    assert hip_lib.mot_host_alloc(C.c_size_t(B * stride * 16), C.byref(hp)) == 0 and hip_lib.mot_host_alloc(C.c_size_t(B * stride * 12), C.byref(hq)) == 0
    pin4 = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_float)), shape=(B, stride, 4))
    pin3 = np.ctypeslib.as_array(C.cast(hq, C.POINTER(C.c_float)), shape=(B, stride, 3))
    with mot.Context(max_points=stride, max_batch=B, max_tracks_total=256) as a, mot.Context(max_points=stride, max_batch=B, max_tracks_total=256) as b:
        for f in range(6):
            n = [N, N - 1234, 2048, 7]
            host = np.zeros((B, stride, 4), np.float32)
            for s in range(B):
                host[s, : n[s]] = synth.make_cloud(N, 30 + s, f)[: n[s]]
            host[..., 3] = 1.0
            kw = dict(run_tracker=True, timestamps=[1.0e9 + f * 1e5] * B, ego_v=[2.0] * B, ego_yaw=[0.002 * f] * B)
            a.wait_uploads(); b.wait_uploads()
            pin4[:] = host; pin3[:] = host[..., :3]
            a.frames_host(hp.value, stride * 4, n, **kw)
            b.frames_host_xyz(hq.value, stride * 3, n, **kw)
            for s in range(B):
                ga, gb = a.get_ground(s, n_hint=n[s]), b.get_ground(s, n_hint=n[s])
                assert np.array_equal(ga["elevated"].view(np.uint32), gb["elevated"].view(np.uint32)) and np.array_equal(ga["ground"].view(np.uint32), gb["ground"].view(np.uint32))
                assert np.array_equal(ga["mask"], gb["mask"])
                assert np.array_equal(a.get_boxes(s)["boxes"], b.get_boxes(s)["boxes"])
                ta, tb = a.get_tracks(s), b.get_tracks(s)
                assert ta["n"] == tb["n"] and np.array_equal(ta["track_manage"], tb["track_manage"]) and np.array_equal(ta["p"], tb["p"])
            g = oracle.ground_remove(p, host[0, : n[0]]); cl = oracle.cluster(p, g["elevated"])
            bx = oracle.box_fit(p, g["elevated"], cl["grid"], cl["num_cluster"])["boxes"]
            assert np.array_equal(b.get_ground(0, n_hint=n[0])["elevated"].view(np.uint32), g["elevated"].view(np.uint32))
            assert np.array_equal(b.get_clusters(0)["grid"], cl["grid"]) and np.array_equal(b.get_boxes(0)["boxes"], bx)
    assert hip_lib.mot_host_free(hp) == 0 and hip_lib.mot_host_free(hq) == 0


@pytest.mark.parametrize("step,ox,oy,oz,ow", [(16, 0, 4, 8, 12), (32, 0, 4, 8, 16), (22, 10, 2, 6, -1)])
def test_frames_host_pointcloud2_equals_frames_host_and_oracle(mot, hip_lib, oracle, synth, step, ox, oy, oz, ow):
    """mot_frames_host_pointcloud2 on the MI355X: one PointCloud2 payload per stream (aligned and unaligned records) against the float4 entry point and the oracle"""
    from test_emu_api_v2 import pointcloud2_payload
    B, N, stride = 4, 60000, 61440
    p = oracle.params(0)
    hp = C.c_void_p()
    assert hip_lib.mot_host_alloc(C.c_size_t(B * stride * 16), C.byref(hp)) == 0
    pin4 = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_float)), shape=(B, stride, 4))
    with mot.Context(max_points=stride, max_batch=B, max_tracks_total=256) as a, mot.Context(max_points=stride, max_batch=B, max_tracks_total=256) as b:
        for f in range(4):
            n = [N, N - 1234, 2048, 7]
            host = np.zeros((B, stride, 4), np.float32)
            for s in range(B):
                host[s, : n[s]] = synth.make_cloud(N, 30 + s, f)[: n[s]]
            if ow < 0:
                host[..., 3] = 1.0
            kw = dict(run_tracker=True, timestamps=[1.0e9 + f * 1e5] * B, ego_v=[2.0] * B, ego_yaw=[0.002 * f] * B)
            a.wait_uploads(); b.wait_uploads()
            pin4[:] = host
            payloads = [pointcloud2_payload(host[s, : n[s]], step, ox, oy, oz, ow, 5 * f + s) for s in range(B)]
            a.frames_host(hp.value, stride * 4, n, **kw)
            b.frames_host_pointcloud2(payloads, n, step, ox, oy, oz, ow, **kw)
            b.wait_uploads()
            for s in range(B):
                ga, gb = a.get_ground(s, n_hint=n[s]), b.get_ground(s, n_hint=n[s])
                assert np.array_equal(ga["elevated"].view(np.uint32), gb["elevated"].view(np.uint32)) and np.array_equal(ga["ground"].view(np.uint32), gb["ground"].view(np.uint32))
                assert np.array_equal(ga["mask"], gb["mask"]) and np.array_equal(a.get_boxes(s)["boxes"], b.get_boxes(s)["boxes"])
                ta, tb = a.get_tracks(s), b.get_tracks(s)
                assert ta["n"] == tb["n"] and np.array_equal(ta["track_manage"], tb["track_manage"]) and np.array_equal(ta["p"], tb["p"])
            g = oracle.ground_remove(p, host[0, : n[0]]); cl = oracle.cluster(p, g["elevated"])
            bx = oracle.box_fit(p, g["elevated"], cl["grid"], cl["num_cluster"])["boxes"]
            assert np.array_equal(b.get_ground(0, n_hint=n[0])["elevated"].view(np.uint32), g["elevated"].view(np.uint32))
            assert np.array_equal(b.get_clusters(0)["grid"], cl["grid"]) and np.array_equal(b.get_boxes(0)["boxes"], bx)
    assert hip_lib.mot_host_free(hp) == 0


def test_track_steps_dev_equals_track_step(mot, hip_lib):
    B = 8
    rng = np.random.default_rng(5)
    with mot.Context(max_points=1024, max_batch=B, max_tracks_total=512) as a, mot.Context(max_points=1024, max_batch=B, max_tracks_total=512) as b:
        centres = [rng.uniform(-30, 30, size=(40, 2)) for _ in range(B)]
        vel = [rng.uniform(-1, 1, size=(40, 2)) for _ in range(B)]
        stride = 48 * 24
        for f in range(12):
            ts = 1.0e9 + f * 1e5
            blk = np.zeros((B, stride), np.float32); m = []
            for s in range(B):
                k = 40 - (s + f) % 5
                c = centres[s][:k] + vel[s][:k] * (0.1 * f)
                bx = np.zeros((k, 8, 3), np.float32)
                bx[:, :, :2] = c[:, None, :] + np.array([[0, 0], [1.8, 0], [1.8, 0.9], [0, 0.9]] * 2)[None]
                bx[:, :4, 2] = -2.0; bx[:, 4:, 2] = 0.4
                blk[s, : k * 24] = bx.reshape(-1); m.append(k)
                a.ego_update(ts, 0.0, 0.0, s); b.ego_update(ts, 0.0, 0.0, s)
                a.track_step(bx, ts, s)
            dev = hiprt.DeviceBuffer(blk)
            b.track_steps_dev(dev.ptr, stride, m, [ts] * B)
            for s in range(B):
                ta, tb = a.get_tracks(s), b.get_tracks(s)
                assert ta["n"] == tb["n"] and np.array_equal(ta["track_manage"], tb["track_manage"]) and np.array_equal(ta["lifetime"], tb["lifetime"])
                assert np.array_equal(ta["p"], tb["p"]) and np.array_equal(ta["v_yaw"], tb["v_yaw"])
        assert (a.get_tracks(0)["track_manage"] > 0).sum() >= 30    # the tracker really runs with tens of live tracks


def test_launch_graphs_equal_plain_launches(mot, hip_lib, oracle, synth):
    """mot_set_launch_graphs: the fused sequence as one hipGraph launch per frame (captured once per launch geometry) must give what the
    plain launches give — frames at changing addresses, changing point counts (inside one geometry and across two), tracker on, the ground
    cloud materialised on demand afterwards — and what the oracle gives."""
    p = oracle.params(0)
    stride = 32768
    with mot.Context(max_points=stride, max_batch=2, max_tracks_total=128) as a, mot.Context(max_points=stride, max_batch=2, max_tracks_total=128) as g:
        g.set_launch_graphs(True)
        bufs = []
        for f in range(10):
            n = [30000 - 37 * f, 17000 + 501 * f] if f % 4 != 3 else [9000 + f, 30000]     # a second geometry every fourth frame
            host = np.zeros((2, stride, 4), np.float32)
            for s in range(2):
                host[s, : n[s]] = synth.make_cloud(30000, 60 + s, f)[: n[s]]
            dev = hiprt.DeviceBuffer(host); bufs.append(dev)     # a new address every frame
            kw = dict(run_tracker=True, timestamps=[1.0e9 + f * 1e5] * 2, ego_v=[2.0, 1.0], ego_yaw=[0.01 * f, 0.0])
            a.frames_dev(dev.ptr, stride * 4, n, **kw); g.frames_dev(dev.ptr, stride * 4, n, **kw)
            for s in range(2):
                assert np.array_equal(a.get_boxes(s)["boxes"], g.get_boxes(s)["boxes"]), (f, s)
                # what reads the resident elevated cloud AFTER the sequence (12-byte points since round 5; the previous frame's mot_get_ground
                # left float4 records and the host-side layout flag with them: a graph REPLAY must set it again, it does not pass the launch code)
                assert np.array_equal(a.box_markers(s).view(np.uint32), g.box_markers(s).view(np.uint32)), (f, s)
                pa, pg = a.cluster_products(s), g.cluster_products(s)
                assert all(np.array_equal(pa[k], pg[k]) for k in ("clustered", "obstacles", "cost_map")), (f, s)
                if f % 2:   # the elevated cloud alone (no ground / mask with it)
                    ea = a.get_ground(s, n_hint=n[s]); eg = np.empty((stride, 4), np.float32); ne = C.c_int(0)
                    assert hip_lib.mot_get_ground(g._h, s, eg.ctypes.data_as(C.c_void_p), C.byref(ne), None, None, None, stride) == 0
                    assert ne.value == ea["n_elevated"] and np.array_equal(eg[: ne.value], ea["elevated"]), (f, s)
                ta, tg = a.get_tracks(s), g.get_tracks(s)
                assert ta["n"] == tg["n"] and np.array_equal(ta["track_manage"], tg["track_manage"]) and np.array_equal(ta["p"], tg["p"]) and np.array_equal(ta["v_yaw"], tg["v_yaw"]), (f, s)
                ga, gg = a.get_ground(s, n_hint=n[s]), g.get_ground(s, n_hint=n[s])
                assert np.array_equal(ga["mask"], gg["mask"]) and np.array_equal(ga["ground"], gg["ground"]) and np.array_equal(ga["elevated"], gg["elevated"])
            o = oracle.ground_remove(p, host[0, : n[0]])
            assert np.array_equal(g.get_ground(0, n_hint=n[0])["elevated"], o["elevated"])
            cl = oracle.cluster(p, o["elevated"])
            assert np.array_equal(g.get_boxes(0)["boxes"], oracle.box_fit(p, o["elevated"], cl["grid"], cl["num_cluster"])["boxes"])
        for d in bufs:
            d.free()


def test_by_products_on_demand_equal_materialised(mot, hip_lib, oracle, synth):
    """mot_set_fused_outputs on the device: ground cloud, mask and per-point cluster labels written by the fused path or computed
    when asked for (compaction re-run; labels from the handed-over cells, or from the points on a 256-cell grid and after a
    stage-wise mot_cluster) — every way equal to the oracle, ragged batch"""
    B, N, stride = 3, 60000, 61440
    n = [N, N - 4321, 900]
    host = np.zeros((B, stride, 4), np.float32)
    for s in range(B):
        host[s, : n[s]] = synth.make_cloud(N, 90 + s, 3)[: n[s]]
    dev = hiprt.DeviceBuffer(host)
    for num_grid in (250, 256):
        po = oracle.params(0, num_grid=num_grid)
        want = []
        for s in range(B):
            g = oracle.ground_remove(po, host[s, : n[s]]); want.append((g, oracle.cluster(po, g["elevated"])))
        for flags in (0, mot.OUT_LABELS, mot.OUT_GROUND | mot.OUT_MASK | mot.OUT_LABELS):
            with mot.Context(mot.params(0, num_grid=num_grid), max_points=stride, max_batch=B) as c:
                c.set_fused_outputs(flags)
                c.frames_dev(dev.ptr, stride * 4, n)
                for s in (2, 0, 1):
                    g, cl = want[s]
                    got = c.get_clusters(s, n_elevated=len(g["elevated"]))
                    assert np.array_equal(got["grid"], cl["grid"]) and np.array_equal(got["point_label"], cl["point_label"]), (num_grid, flags, s)
                    gg = c.get_ground(s, n_hint=n[s])
                    assert np.array_equal(gg["mask"], g["mask"]) and np.array_equal(gg["ground"].view(np.uint32), g["ground"].view(np.uint32))
                    assert np.array_equal(c.get_clusters(s, n_elevated=len(g["elevated"]))["point_label"], cl["point_label"])   # still there after the compaction re-run
    po = oracle.params(0)
    g = oracle.ground_remove(po, host[0, :N])
    a = np.ascontiguousarray(g["elevated"][::-1])
    ref = oracle.cluster(po, a)
    with mot.Context(max_points=stride, max_batch=B) as c:
        c.frames_dev(dev.ptr, stride * 4, n)
        G = c.params.num_grid
        grid = np.zeros((G, G), np.int32); nc = C.c_int(0)
        assert hip_lib.mot_cluster(c._h, a.ctypes.data_as(C.c_void_p), len(a), grid.ctypes.data_as(C.c_void_p), C.byref(nc), None) == 0
        got = c.get_clusters(0, n_elevated=len(a))
        assert np.array_equal(got["grid"], ref["grid"]) and np.array_equal(got["point_label"], ref["point_label"])


def test_stage_wise_call_after_a_fused_batch_leaves_the_other_slots_readable(mot, hip_lib, oracle, synth):
    """tests/mixed_use_case.py on the device: the fused batch's 12-byte elevated points in slots 1, 2 stay readable (labels, side products, cubes)
    after a stage-wise call has put float4 records into slot 0, and the node-frame call after a fused batch reads its own upload as float4"""
    import mixed_use_case
    bufs = []
    def upload(host):
        bufs.append(hiprt.DeviceBuffer(host)); return bufs[-1].ptr
    try:
        mixed_use_case.run(mot, None, synth, oracle, upload=upload, N=40000)
        mixed_use_case.run_ground_after_takeover(mot, None, synth, oracle, upload=upload, N=40000)
    finally:
        for d in bufs: d.free()


def test_trace_ranges_on_the_device(mot, hip_lib, synth):
    """roctx stage ranges (mot_set_trace_ranges) change nothing but the markers: a frame with them on equals a frame with them off"""
    cloud = synth.make_cloud(30000, 2, 0)
    out = []
    for on in (False, True):
        with mot.Context(max_points=32768, max_tracks_total=64) as c:
            if on:
                rc = c.lib.mot_set_trace_ranges(c._h, 1)
                assert rc in (mot.MOT_OK, mot.MOT_E_STATE)   # E_STATE: no libroctx64 on this box (ranges stay off)
            dev = hiprt.DeviceBuffer(np.pad(cloud, ((0, 32768 - len(cloud)), (0, 0))))
            c.frames_dev(dev.ptr, 32768 * 4, [len(cloud)], run_tracker=True, timestamps=[1e9], ego_v=[0.0], ego_yaw=[0.0])
            out.append((c.get_boxes(0)["boxes"], c.get_clusters(0)["grid"]))
            dev.free()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_stream_snapshot_round_trip(mot, hip_lib):
    """mot_stream_save / mot_stream_load on the MI355X: a stream moved to another slot of another context continues bit for bit
    (every output and filter state of every later frame; tests/snapshot_case.py)"""
    import snapshot_case
    snapshot_case.check(mot)
