"""mot_ring (include/mot.h): contexts of one device joined into a stage ring. The ring only ORDERS the stages of different
contexts by HIP events; every result must be what free-running contexts produce. Here: the emulator build (tests/emu, events
are trivial there, so this checks the bookkeeping — membership, re-ringing, dissolve, destroy, argument errors). The same
comparison on the MI355X, where the events really order concurrent streams: tests/test_ring_gpu.py."""
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


def run_ring_case(mot, synth, lib_path, stages, n_ctx, B, N, stride, frames, sync_each=False):
    """n_ctx ring members and one free context get the same frames; every member must end up with the free context's state"""
    mk = lambda: mot.Context(max_points=stride, max_batch=B, max_tracks_total=256, **({"lib_path": lib_path} if lib_path else {}))
    ring, free = [mk() for _ in range(n_ctx)], mk()
    try:
        mot.ring(ring, stages)
        for f in range(frames):
            n = [N - 311 * s for s in range(B)]
            host = np.zeros((B, stride, 4), np.float32)
            for s in range(B):
                host[s, : n[s]] = synth.make_cloud(N, 70 + s, f)[: n[s]]
            kw = dict(run_tracker=True, timestamps=[1.0e9 + f * 1e5] * B, ego_v=[2.0] * B, ego_yaw=[0.02 * f] * B)
            for c in ring:   # ring order
                c.frames_host(host.ctypes.data, stride * 4, n, **kw)
            free.frames_host(host.ctypes.data, stride * 4, n, **kw)
            for c in ring + [free]:
                c.wait_uploads()
                if sync_each:
                    c.synchronize()
        for c in ring:
            for s in range(B):
                ga, gb = c.get_ground(s, n_hint=N), free.get_ground(s, n_hint=N)
                assert np.array_equal(ga["elevated"], gb["elevated"]) and np.array_equal(ga["mask"], gb["mask"])
                ca, cb = c.get_clusters(s), free.get_clusters(s)
                assert ca["num_cluster"] == cb["num_cluster"] and np.array_equal(ca["grid"], cb["grid"])
                assert np.array_equal(c.get_boxes(s)["boxes"], free.get_boxes(s)["boxes"])
                ta, tb = c.get_tracks(s), free.get_tracks(s)
                assert ta["n"] == tb["n"] and np.array_equal(ta["track_manage"], tb["track_manage"])
                assert np.array_equal(ta["p"], tb["p"]) and np.array_equal(ta["v_yaw"], tb["v_yaw"]) and np.array_equal(ta["vis_box"], tb["vis_box"])
    finally:
        for c in ring + [free]:
            c.close()


@pytest.mark.parametrize("stages,n_ctx", [(2, 2), (3, 3), (4, 2)])
def test_ring_results_unchanged(mot, emu, synth, stages, n_ctx):
    run_ring_case(mot, synth, emu[0], stages, n_ctx, B=2, N=4000, stride=4096, frames=3)


def test_ring_membership(mot, emu):
    lib, L = emu
    mk = lambda: mot.Context(lib_path=lib, max_points=1024, max_batch=1)
    a, b, c = mk(), mk(), mk()
    arr = lambda *cs: (C.c_void_p * len(cs))(*[x._h for x in cs])
    assert L.mot_ring(arr(a, b), 2, 5) != 0 and b"stages" in L.mot_last_error(a._h)      # 2, 3 or 4
    assert L.mot_ring(arr(a, a), 2, 2) != 0                                               # a context twice
    assert L.mot_ring(None, 2, 2) != 0
    mot.ring([a, b, c], 3)
    mot.ring([a, b], 2)          # re-ringing dissolves the old ring of all three first
    pts = np.zeros((1, 1024, 4), np.float32); pts[0, :, 0] = np.linspace(5, 20, 1024); pts[0, :, 2] = -1.0
    for x in (a, b, c):          # c is free again: its stages wait for nobody
        x.frames_host(pts.ctypes.data, 1024 * 4, [1024]); x.wait_uploads(); x.synchronize()
    b.close()                    # destroying a member dissolves the ring: a must not touch b's events afterwards
    a.frames_host(pts.ctypes.data, 1024 * 4, [1024]); a.wait_uploads(); a.synchronize()
    mot.ring([a, c], 0)          # dissolve explicitly (no-op here)
    a.close(); c.close()
