"""include/mot.h mot_gather_* (ABI v6: the per-tick all-gather of the live-track blocks issued from C) on the emulator build of the kernels — one rank, no
communicator (the tick's collective is a copy), a host thread PER CONTEXT contributing as bench.py's issuing threads do: every tick's receive buffer must equal
what multi.TrackGatherAll (the torch / Python shim it replaces in the data loop) produces from the same contexts, and what mot_get_tracks reports as live.
The rendezvous logic (ticks, double buffering, the thread that completes a tick issues the collective) is what this covers on CPU; tests/test_gather_gpu.py runs
the same comparison on the MI355X with a one-rank RCCL communicator."""
import ctypes as C
import os
import sys
import threading

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
REC = np.dtype([("id", "i4"), ("track_manage", "i4"), ("is_static", "i4"), ("is_vis", "i4"), ("p", "f4", 3), ("lifetime", "i4"), ("v_yaw", "f8", 2), ("vis_box", "f4", 24)])


def boxes(f, k):
    b = np.zeros((k, 8, 3), np.float32)
    for i in range(k):
        b[i, :, :2] = np.array([[0, 0], [2, 0], [2, 1], [0, 1]] * 2) + [6.0 * i - 15 + 0.3 * f, 4.0 + 0.1 * f * (i % 3)]
        b[i, :4, 2] = -2.0; b[i, 4:, 2] = 0.5
    return b


def decode(raw, world, nc, block, batch, cap):
    head = (batch * 4 + 15) & ~15
    out = []
    for r in range(world):
        per = []
        for ci in range(nc):
            blk = raw[(r * nc + ci) * block:(r * nc + ci + 1) * block]
            counts = blk[: batch * 4].view(np.int32).copy()
            recs = blk[head: head + min(int(counts.sum()), cap) * REC.itemsize].view(REC).copy()
            per.append((counts, recs))
        out.append(per)
    return out


def test_native_gather_equals_the_python_shim_thread_per_context(mot):
    import build_emu
    from conftest import load_sub
    multi = load_sub("multi")
    lib = build_emu.build()
    NC, B, CAP, F = 4, 3, 3 * 8, 14
    ctxs = [mot.Context(lib_path=lib, max_points=1024, max_batch=B, max_tracks_total=64) for _ in range(NC)]
    try:
        with mot.NativeGather(ctxs, B, CAP) as g:
            shim = multi.TrackGatherAll(ctxs, B, CAP, 1, "cpu")
            assert g.block == shim.block
            seen = {}
            barrier = threading.Barrier(NC + 1)
            emu_lock = threading.Lock()   # the emulator runs a kernel on the calling thread and is not re-entrant: library calls one at a time (the threads still
            errs = []                     # reach contribute() in any order, and whoever completes the tick issues the collective)

            def feed(ci):
                try:
                    for f in range(F):
                        cx = ctxs[ci]
                        ts = 1.0e9 + f * 1e5
                        bx = np.stack([boxes(f + s, 4 + ci) for s in range(B)]).reshape(B, -1)
                        with emu_lock:
                            for s in range(B):
                                cx.ego_update(ts, 1.0, 0.002 * f, s)
                            cx.track_steps_dev(bx.ctypes.data, bx.shape[1], [4 + ci] * B, [ts] * B)
                            g.contribute(ci)
                        barrier.wait()   # the main thread reads this tick's result ...
                        barrier.wait()   # ... before anybody starts the tick after next (which rewrites its buffer)
                except BaseException as e:
                    errs.append(e); barrier.abort()

            th = [threading.Thread(target=feed, args=(ci,)) for ci in range(NC)]
            for t in th:
                t.start()
            for f in range(F):
                barrier.wait()
                assert not errs, errs
                d, nb, tick, _ev = g.result()
                g.synchronize()
                assert tick == f + 1 and nb == g.block
                raw = np.ctypeslib.as_array(C.cast(d, C.POINTER(C.c_uint8)), shape=(NC * nb,)).copy()
                ref = shim.step().numpy().copy()   # the shim on the same contexts, same state: the same bytes
                assert np.array_equal(raw, ref[: len(raw)]), f
                seen[f] = decode(raw, 1, NC, nb, B, CAP)
                barrier.wait()
            for t in th:
                t.join()
            assert not errs, errs
            # the last tick against mot_get_tracks
            for ci, cx in enumerate(ctxs):
                counts, recs = seen[F - 1][0][ci]
                off = np.concatenate([[0], np.cumsum(counts)])
                for s in range(B):
                    t = cx.get_tracks(s)
                    live = np.nonzero(t["track_manage"] > 0)[0]
                    assert counts[s] == len(live) > 0 and np.array_equal(recs["id"][off[s]:off[s + 1]], live)
                    assert np.array_equal(recs["p"][off[s]:off[s + 1]], t["p"][live])
    finally:
        for cx in ctxs:
            cx.close()


def test_native_gather_arguments(mot):
    import build_emu
    lib = build_emu.build()
    with mot.Context(lib_path=lib, max_points=1024, max_batch=2, max_tracks_total=64) as cx:
        L = cx.lib
        g = C.c_void_p()
        arr = (C.c_void_p * 1)(cx._h)
        assert L.mot_gather_create(arr, 1, 3, 8, 1, 0, None, C.byref(g)) == mot.MOT_E_ARG       # more streams than the context has slots
        assert L.mot_gather_create(arr, 1, 2, 8, 2, 0, None, C.byref(g)) == mot.MOT_E_ARG       # two ranks need a unique id
        assert L.mot_gather_create(arr, 1, 2, 8, 2, 0, (C.c_char * 128)(), C.byref(g)) == mot.MOT_E_STATE   # ... and RCCL: none in the emulator build
        assert L.mot_gather_create(arr, 1, 2, 8, 1, 0, None, C.byref(g)) == mot.MOT_OK
        assert L.mot_gather_result(g, None, None, None, None) == mot.MOT_E_STATE                 # no tick completed yet
        assert L.mot_gather_contribute(g, 1) == mot.MOT_E_ARG
        assert L.mot_gather_destroy(g) == mot.MOT_OK
