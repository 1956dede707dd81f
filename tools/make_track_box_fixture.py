"""Packs the box streams of tools/dump_track_streams.py dumps (gpurun_out/*.npz, made on the GPU box: the renderer is csrc/synth.hip) into
tests/golden/track_boxes.npz — per stream the boxes in the global frame that the tracker was fed, frame by frame. These are the reference's
own boxes (tests/test_sequence_gpu.py: sensor-frame and global-frame boxes of these streams are bit-equal to oracle/_ref's), so the fixture
is "what the reference's tracker sees on the bench / test streams"; tests/test_tracker_noise_floor.py replays it on CPU.
  python tools/make_track_box_fixture.py gpurun_out/bench120.npz gpurun_out/units01.npz gpurun_out/pts200k.npz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, names = {}, []
for path in sys.argv[1:]:
    d = np.load(path)
    F = len(d["ego_v"])
    for b, scene in enumerate(d["scenes"]):
        name = f"scene{int(scene)}_{int(d['points']) // 1000}k_unit{float(d['unit']):g}_preset{int(d['preset'])}"
        nb = [len(d[f"s{b}_f{f}_boxes_global"]) for f in range(F)]
        out[name + "/n_boxes"] = np.array(nb, np.int32)
        out[name + "/boxes_global"] = np.concatenate([d[f"s{b}_f{f}_boxes_global"] for f in range(F)]).astype(np.float32)
        out[name + "/meta"] = np.array([int(scene), int(d["points"]), float(d["unit"]), int(d["preset"]), F], np.float64)
        out[name + "/ego_v"] = d["ego_v"]; out[name + "/ego_yaw"] = d["ego_yaw"]
        names.append(name)
out["streams"] = np.array(names)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "track_boxes.npz"), **out)
print("tests/golden/track_boxes.npz:", names)
