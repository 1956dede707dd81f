"""The adapter header's tracker functions (include/mot_adapters.hpp: getOriginPoints + immUkfJpdaf with the reference's signatures) on a
track budget far smaller than the run needs — shared by the emulator test (tests/test_adapters_run.py) and the -m gpu test
(tests/test_nodes_gpu.py). The round-4 review's finding: the adapter read back into a fixed buffer of max_tracks_total records and threw
for ever once a stream had CREATED more tracks than that (the reference's outputs have one entry per track ever created,
OT/tracking/imm_ukf_jpda.cpp:995-1041, and simply grow)."""
import struct
import subprocess

import numpy as np

import tracker_cases as TC


def write_boxes(path, world):
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(world)))
        for boxes, ts, v, yaw in world:
            b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 8, 3)
            f.write(struct.pack("<iddd", len(b), ts, v, yaw)); f.write(b.tobytes())


def read_out(path):
    frames = []
    for line in open(path):
        w = line.split()
        recs = [tuple(float(x) for x in r.split(":")) for r in w[5:]]
        frames.append(dict(n=int(w[1]), n_vis=int(w[2]), ego=(float(w[3]), float(w[4])), recs=np.array(recs, np.float64).reshape(-1, 7)))
    return frames


def run(driver, oracle, tmp_path, slots=24, ever=150, frames=420, seed=5, spots=9, rtol=1e-4):
    """-> number of restarts. Until the stream has created `ever` tracks the adapter must return what the oracle's tracker returns with
    unbounded memory (its record buffer starts at `slots` and has to grow); when the budget is used up it must say so, restart the
    stream's tracks and go on — never throw."""
    world = list(TC.blinking_world(seed, spots, frames))
    i, o = str(tmp_path / "boxes.bin"), str(tmp_path / "out.txt")
    write_boxes(i, world)
    r = subprocess.run([driver, i, o, str(slots), str(ever)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    out = read_out(o)
    assert len(out) == frames
    restarts = r.stderr.count("restarting the tracks of this stream")
    p = oracle.params(0)
    T = oracle.Tracker(p)
    compared = grown = 0
    for f, (boxes, ts, v, yaw) in enumerate(world):
        e = T.ego_update(ts, v, yaw)
        ref = T.step(boxes, ts, max_tracks=1 << 14)
        a = out[f]
        assert np.allclose(a["ego"], e[:2], rtol=1e-12, atol=1e-12), f
        if ref["n"] >= ever:     # the budget is reached in this frame: from here on the adapter runs restarted tracks
            break
        assert a["n"] == ref["n"], (f, a["n"], ref["n"])
        assert np.array_equal(a["recs"][:, 0], ref["track_manage"]) and np.array_equal(a["recs"][:, 1], ref["is_static"]) and np.array_equal(a["recs"][:, 2], ref["is_vis"]), f
        live = ref["track_manage"] > 0
        assert np.allclose(a["recs"][live, 3:5], ref["p"][live, :2], rtol=rtol, atol=1e-4, equal_nan=True), f   # (a diverged track is NaN on both sides)
        assert a["n_vis"] == int(ref["is_vis"].sum()), f
        compared += 1; grown = max(grown, a["n"])
    T.close()
    assert grown > 2 * slots, grown          # far more records than the initial buffer: it grew
    assert compared > 100 and compared < frames, compared
    assert restarts >= 1, r.stderr[-500:]     # ... and the budget WAS used up later in the run
    assert all(x["n"] <= ever for x in out)
    assert max(x["n"] for x in out[-20:]) > 0   # still tracking at the end
    return restarts


def run_refused(driver, tmp_path):
    """round-5 advice: a frame with more boxes than the library takes (MOT_MAX_BOXES_PER_FRAME) is refused before the step runs — the adapter must THROW
    for it (the reference's own immUkfJpdaf would have taken it: an error of this call, told loudly) and must NOT read it as "births dropped" and
    wipe the stream's tracks, as it did while both came back as a bare MOT_E_CAPACITY."""
    world = list(TC.blinking_world(3, 4, 12))
    boxes, ts, v, yaw = world[-1]
    one = np.zeros((1, 8, 3), np.float32); one[0, :, :2] = [[0, 0], [2, 0], [2, 1], [0, 1]] * 2; one[0, 4:, 2] = 1.0
    world.append((np.concatenate([one + [3.0 * (k % 40), 3.0 * (k // 40), 0] for k in range(1025)]), ts + 1e5, v, yaw))
    i, o = str(tmp_path / "boxes_refused.bin"), str(tmp_path / "out_refused.txt")
    write_boxes(i, world)
    r = subprocess.run([driver, i, o, "64", "4096"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "more boxes in a frame" in r.stderr and "restarting the tracks" not in r.stderr, (r.returncode, r.stderr[-800:])
