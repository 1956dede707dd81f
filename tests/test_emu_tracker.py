"""CPU-only development check of the tracker kernel's LOGIC under tests/emu/hipemu.h against the golden vectors
(see hipemu.h: not a product path, not a parity claim)."""
import os
import sys

import numpy as np
import pytest

import golden_util as G

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.mark.parametrize("name", G.TRACKERS + G.TRACKERS_OT0)
def test_emu_tracker_golden(mot, name):
    import build_emu
    lib = build_emu.build()
    fx = G.load(name)
    ot0 = name in G.TRACKERS_OT0   # object_tracking0's tracker: preset 1; boxes are only shown after lifetime 8
    with mot.Context(mot.params(int(ot0), lib=mot.load_library(lib)), lib_path=lib, max_points=4096, max_tracks_total=256) as c:
        for f in range(18 if ot0 else 14):
            ts = 1.0e9 + f * float(fx["unit"])
            ego = c.ego_update(ts, *G.ego_of(fx, f))
            assert np.allclose(ego, fx["ego"][f], rtol=1e-12, atol=1e-12)
            out = c.track_step(fx["boxes"][f][: fx["n_boxes"][f]], ts)
            G.check_tracker_frame(fx, f, out, lambda i: c.track_state(i), rtol=1e-6)


def test_sequence_player_kitti_layout(mot, oracle, synth, tmp_path):
    """the rosbag-free player (sequence.py) on a KITTI-raw-shaped directory of synthetic scans: every frame equals the oracle's
    ground -> cluster -> box -> tracker chain (kernels emulated here; test_tracker_gpu.py runs the same check on the GPU)"""
    import build_emu
    import conftest
    seq = conftest.load_sub("sequence")
    lib = build_emu.build()
    d = tmp_path / "2011_09_26_drive_0005_sync"
    (d / "velodyne_points" / "data").mkdir(parents=True); (d / "oxts" / "data").mkdir(parents=True)
    for f in range(4):
        synth.make_cloud(12000, 3, f).tofile(d / "velodyne_points" / "data" / f"{f:010d}.bin")
        ox = np.zeros(30); ox[8] = 1.5 + 0.1 * f; ox[5] = 0.002 * f
        np.savetxt(d / "oxts" / "data" / f"{f:010d}.txt", ox[None])
    p = oracle.params(0)
    T = oracle.Tracker(p)
    with mot.Context(lib_path=lib, max_points=16384, max_tracks_total=256) as c:
        n = 0
        for k, r in enumerate(seq.play(c, seq.kitti_frames(str(d)))):
            cloud = synth.make_cloud(12000, 3, k); ts = 1.0e9 + k * 1.0e5
            g = oracle.ground_remove(p, cloud); cl = oracle.cluster(p, g["elevated"])
            bx = oracle.box_fit(p, g["elevated"], cl["grid"], cl["num_cluster"])["boxes"]
            assert r["n_elevated"] == len(g["elevated"]) and np.array_equal(r["boxes"], bx)
            ego = T.ego_update(ts, 1.5 + 0.1 * k, 0.002 * k)
            assert np.allclose(r["ego"], ego, rtol=1e-12, atol=1e-12)
            o = T.step(seq.boxes_to_global(bx, ego), ts)
            assert r["tracks"]["n"] == o["n"] and np.array_equal(r["tracks"]["track_manage"], o["track_manage"])
            n += 1
        assert n == 4
    T.close()


def test_fused_sequence_player_equals_the_stagewise_one(mot, oracle, synth):
    """sequence.play_fused (one frames_host call per frame, two streams) against sequence.play (the nodes' stage-wise calls) and the
    oracle fed through the tracking node's tf chain: boxes bit-exact, track sets / trackManage equal, positions within 1e-4"""
    import build_emu
    import conftest
    import seq_parity as SP
    seq = conftest.load_sub("sequence")
    lib = build_emu.build()
    F, N = 6, 12000
    frames = [[(synth.make_cloud(N, 3 + s, f), 1.5 + 0.1 * f, 0.002 * f * (1 + s)) for s in range(2)] for f in range(F)]
    p = oracle.params(0)
    with mot.Context(lib_path=lib, max_points=12288, max_batch=2, max_tracks_total=256) as c:
        fused = list(seq.play_fused(c, frames, slot_count=2))
    assert len(fused) == F
    for s in range(2):
        T = oracle.Tracker(p)
        with mot.Context(lib_path=lib, max_points=12288, max_tracks_total=256) as c1:
            for f, r in enumerate(seq.play(c1, [fr[s] for fr in frames])):
                got = fused[f][s]
                assert np.array_equal(got["boxes"], r["boxes"]), (f, s)
                ts = 1.0e9 + f * 1.0e5
                ego = T.ego_update(ts, frames[f][s][1], frames[f][s][2])
                o = T.step(SP.boxes_to_global(oracle, c1.lib, r["boxes"], ego), ts)
                assert got["tracks"]["n"] == o["n"] and np.array_equal(got["tracks"]["track_manage"], o["track_manage"]), (f, s)
                live = o["track_manage"] > 0
                assert np.allclose(got["tracks"]["p"][live], o["p"][live], rtol=1e-4, atol=1e-5), (f, s)
        T.close()


def test_tracker_launch_modes_give_identical_results(mot, oracle):
    """mot_set_tracker_mode: the step as four launches (tracks of all streams dealt over the chip — what the bench's 512-stream contexts run)
    and as ONE launch with a workgroup per stream (what contexts of few streams run) must give the same bits: same phases, same order per
    track, kernel boundaries replaced by workgroup barriers. Random and degenerate sequences on three streams at once, 40 live tracks."""
    import build_emu
    import seq_parity as SP
    import test_emu_tracker_random as TR
    import tracker_cases as TC
    lib = build_emu.build()
    results = {}
    for mode in (1, 2):
        res = []
        with mot.Context(lib_path=lib, max_points=1024, max_batch=3, max_tracks_total=256) as c:
            c.set_tracker_mode(mode)
            seqs = [TR.sequence(70 + k, frames=20) if k < 2 else TR.hostile_sequence(5, frames=20) for k in range(3)]
            for f in range(20):
                for s in range(3):
                    boxes, ts, v, yaw = seqs[s][f]
                    c.ego_update(ts, v, yaw, s)
                    tr = c.track_step(boxes, ts, s)
                    res.append((tr, {int(i): c.track_state(int(i), slot=s) for i in np.nonzero(tr["track_manage"] > 0)[0]}))
            # and the batched device entry point with many live tracks per stream (more than one round of 32 groups in the stream kernel)
            st = None
        results[mode] = res
        st = TC.many_live_tracks(_ModeCtx(mot, mode), oracle, lambda a: (a.ctypes.data, lambda: None), lib_path=lib, streams=2, T=40, frames=14, spacing=9.0, min_live=40)
        results[(mode, "many")] = st["max_rel_state_err"]
    for (ta, sa), (tb_, sb) in zip(results[1], results[2]):
        bits = lambda a: np.ascontiguousarray(a).view(np.uint8)   # (NaN outputs of a diverged track must be the same NaNs)
        assert ta["n"] == tb_["n"] and all(np.array_equal(bits(ta[k]), bits(tb_[k])) for k in ("track_manage", "lifetime", "is_static", "is_vis", "p", "v_yaw", "vis_box"))
        assert sa.keys() == sb.keys()
        for i in sa:
            for k in SP.STATE_KEYS:
                assert np.array_equal(bits(np.asarray(sa[i][k])), bits(np.asarray(sb[i][k]))), (i, k)
    assert results[(1, "many")] == results[(2, "many")]


class _ModeCtx:
    """the package with every new Context put into one tracker launch mode (for helpers that create their own contexts)"""

    def __init__(self, mot, mode):
        self._mot, self._mode = mot, mode

    def __getattr__(self, k):
        return getattr(self._mot, k)

    def Context(self, *a, **kw):
        c = self._mot.Context(*a, **kw); c.set_tracker_mode(self._mode); return c


def test_sum_order_knob():
    """-DMOT_TRACK_SEQ_SUMS=1 (mot_wave.h) puts the sums over the sigma points back into the reference's order (ukf.cpp:736-749: explicit
    loops, i = 0..14); the product adds the same terms as a tree over the DPP row. On the golden fixtures (the reference build's own values)
    both stay far inside the bar — and the sequential build is the closer one, which is what makes the knob useful: a parity difference that
    survives it is not reordering noise."""
    import subprocess
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sum_order_probe.py")
    res = {}
    for tag, defs in (("tree", ""), ("seq", "-DMOT_TRACK_SEQ_SUMS=1")):
        env = dict(os.environ, MOT_EMU_DEFINES=defs)
        r = subprocess.run([sys.executable, probe], capture_output=True, text=True, env=env, timeout=1500)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = {ln.split()[0]: float(ln.split()[1]) for ln in r.stdout.splitlines() if ln.strip()}
    print(res)
    assert set(res["tree"]) == set(res["seq"]) and res["tree"]
    for name in res["tree"]:
        assert res["tree"][name] <= 1e-6 and res["seq"][name] <= 1e-6, res
    assert sum(res["seq"].values()) <= sum(res["tree"].values()), res


def test_dense_update_instantiation_is_chosen_by_the_live_track_count():
    """track_update_kernel / track_update_dense_kernel (round 6: the same code at 2 / 3 waves per SIMD; both are launched, the live-track count of the launch decides
    on the device which one works): with the threshold pulled down to 4 tracks (-DMOT_UPDATE_DENSE_TRACKS=4) the golden sequences cross it back and forth —
    frames below it go through the plain kernel, frames above through the dense one, every track through exactly one: the results are those of the default build,
    to the last bit of the probe's figures."""
    import subprocess
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sum_order_probe.py")
    res = {}
    for tag, defs in (("default", ""), ("dense from 4 tracks", "-DMOT_UPDATE_DENSE_TRACKS=4")):
        r = subprocess.run([sys.executable, probe], capture_output=True, text=True, env=dict(os.environ, MOT_EMU_DEFINES=defs), timeout=1500)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = {ln.split()[0]: float(ln.split()[1]) for ln in r.stdout.splitlines() if ln.strip()}
    assert res["default"] and res["default"] == res["dense from 4 tracks"], res
