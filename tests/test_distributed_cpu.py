"""N > 1 harness logic on CPU: 2 processes, gloo, 127.0.0.1. Each rank drives its own streams through the (emulated —
see tests/emu/hipemu.h) kernels, exports its live-track block and all-gathers; every rank must end up with every
rank's block, equal to what the oracle computes for those streams. Covers: stream sharding, export, the collective."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "emu"))


def _worker(rank, world, port, lib, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from conftest import load_pkg, load_sub
        import oracle_lib as O
        import seq_parity as SP
        mot = load_pkg(); synth = load_sub("synth"); multi = load_sub("multi")
        B, N, stride, K = 2, 6000, 6144, 16
        p = O.params(0)
        ctx = mot.Context(lib_path=lib, max_points=stride, max_batch=B, max_tracks_total=256)
        tg = multi.TrackGather(B, K, world, "cpu")
        expect = {}
        trackers = {(r, b): O.Tracker(p) for r in range(world) for b in range(B)}
        for f in range(4):
            host = np.zeros((B, stride, 4), np.float32)
            for b in range(B):
                host[b, :N] = synth.make_cloud(N, multi.scene_of(rank, b), f)
            ts = [1.0e9 + f * 1e5] * B
            ctx.frames_dev(host.ctypes.data, stride * 4, [N] * B, run_tracker=True, timestamps=ts, ego_v=[1.5] * B, ego_yaw=[0.004 * f] * B)
            tg.step(ctx)
            # oracle for EVERY rank's streams (each rank checks the whole gathered result)
            for r in range(world):
                for b in range(B):
                    c = synth.make_cloud(N, multi.scene_of(r, b), f)
                    g = O.ground_remove(p, c); cl = O.cluster(p, g["elevated"])
                    bx = O.box_fit(p, g["elevated"], cl["grid"], cl["num_cluster"])["boxes"]
                    T = trackers[(r, b)]
                    ego = T.ego_update(ts[b], 1.5, 0.004 * f)
                    expect[(r, b)] = T.step(SP.boxes_to_global(O, ctx.lib, bx, ego[:3]), ts[b])   # the tracking node's tf step
        blocks = tg.blocks_as_numpy()
        assert len(blocks) == world
        for r, (cnt, rec) in enumerate(blocks):
            for b in range(B):
                o = expect[(r, b)]
                live = np.nonzero(o["track_manage"] > 0)[0][:K]
                assert cnt[b] == len(live), (rank, r, b, cnt[b], len(live))
                assert np.array_equal(rec[b]["id"][: cnt[b]], live)
                assert np.array_equal(rec[b]["track_manage"][: cnt[b]], o["track_manage"][live])
                assert np.allclose(rec[b]["p"][: cnt[b]], o["p"][live], rtol=1e-4, atol=1e-6)     # BASELINE.json's bar
                assert np.allclose(rec[b]["v_yaw"][: cnt[b]], o["v_yaw"][live], rtol=1e-4, atol=1e-7)
                assert np.array_equal(rec[b]["is_vis"][: cnt[b]], o["is_vis"][live]) and np.array_equal(rec[b]["lifetime"][: cnt[b]], o["lifetime"][live])
                assert np.allclose(rec[b]["vis_box"][: cnt[b]], o["vis_box"][live], rtol=1e-4, atol=1e-5)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()[-1500:]))


def _worker_packed(rank, world, port, lib, q):
    """the packed form: TWO contexts per rank, ONE collective per frame for both (multi.TrackGatherAll)"""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from conftest import load_pkg, load_sub
        import oracle_lib as O
        import seq_parity as SP
        mot = load_pkg(); synth = load_sub("synth"); multi = load_sub("multi")
        NC, B, N, stride = 2, 2, 5000, 5120
        p = O.params(0)
        ctxs = [mot.Context(lib_path=lib, max_points=stride, max_batch=B, max_tracks_total=256) for _ in range(NC)]
        tg = multi.TrackGatherAll(ctxs, B, 64, world, "cpu")
        small = multi.TrackGatherAll(ctxs, B, 1, world, "cpu")   # a block too small on purpose: truncation must be reported, the counts stay true
        trackers = {(r, ci, b): O.Tracker(p) for r in range(world) for ci in range(NC) for b in range(B)}
        expect = {}
        for f in range(5):
            ts = [1.0e9 + f * 1e5] * B
            for ci, cx in enumerate(ctxs):
                host = np.zeros((B, stride, 4), np.float32)
                for b in range(B):
                    host[b, :N] = synth.make_cloud(N, multi.scene_of(rank, ci * B + b), f)
                cx.frames_dev(host.ctypes.data, stride * 4, [N] * B, run_tracker=True, timestamps=ts, ego_v=[1.5] * B, ego_yaw=[0.004 * f] * B)
            tg.step(); small.step()
            for r in range(world):
                for ci in range(NC):
                    for b in range(B):
                        c = synth.make_cloud(N, multi.scene_of(r, ci * B + b), f)
                        g = O.ground_remove(p, c); cl = O.cluster(p, g["elevated"])
                        bx = O.box_fit(p, g["elevated"], cl["grid"], cl["num_cluster"])["boxes"]
                        T = trackers[(r, ci, b)]
                        ego = T.ego_update(ts[b], 1.5, 0.004 * f)
                        expect[(r, ci, b)] = T.step(SP.boxes_to_global(O, ctxs[0].lib, bx, ego[:3]), ts[b])
        blocks = tg.blocks_as_numpy(); cut = small.blocks_as_numpy()
        assert len(blocks) == world and all(len(per) == NC for per in blocks)
        some_cut = False
        for r in range(world):
            for ci in range(NC):
                counts, recs, trunc = blocks[r][ci]
                assert not trunc
                c2, r2, t2 = cut[r][ci]
                assert np.array_equal(c2, counts) and t2 == (int(counts.sum()) > 1)   # the header always carries the true counts
                some_cut |= int(counts.sum()) > 0
                for b in range(B):
                    o = expect[(r, ci, b)]
                    live = np.nonzero(o["track_manage"] > 0)[0]
                    assert counts[b] == len(live) == len(recs[b]), (rank, r, ci, b, counts[b], len(live))
                    assert np.array_equal(recs[b]["id"], live) and np.array_equal(recs[b]["track_manage"], o["track_manage"][live])
                    assert np.array_equal(recs[b]["lifetime"], o["lifetime"][live]) and np.array_equal(recs[b]["is_vis"], o["is_vis"][live])
                    assert np.allclose(recs[b]["p"], o["p"][live], rtol=1e-4, atol=1e-6) and np.allclose(recs[b]["v_yaw"], o["v_yaw"][live], rtol=1e-4, atol=1e-7)
        assert some_cut
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()[-1500:]))


def test_two_rank_packed_gather_one_collective_for_all_contexts():
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    import build_emu
    lib = build_emu.build()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 30100 + os.getpid() % 500
    procs = [ctxm.Process(target=_worker_packed, args=(r, 2, port, lib, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=600) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_two_rank_stream_sharding_and_track_gather():
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    import build_emu
    lib = build_emu.build()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29500 + os.getpid() % 500
    procs = [ctxm.Process(target=_worker, args=(r, 2, port, lib, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=600) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
