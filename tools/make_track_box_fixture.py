"""Packs the box streams of tools/dump_track_streams.py dumps (gpurun_out/*.npz, made on the GPU box: the renderer is csrc/synth.hip) into
tests/golden/track_boxes.npz — per stream the boxes in the global frame that the tracker was fed, frame by frame. These are the reference's
own boxes (tests/test_sequence_gpu.py: sensor-frame and global-frame boxes of these streams are bit-equal to oracle/_ref's), so the fixture
is "what the reference's tracker sees on the bench / test streams"; tests/test_tracker_noise_floor.py replays it on CPU.
  python tools/make_track_box_fixture.py gpurun_out/bench120.npz gpurun_out/units01.npz gpurun_out/pts200k.npz
Round 6 — the dense (plaza) scene of bench.py's dense_scene leg, streams 7000 and 7001 (MOT_DUMP_SCENE_KIND=plaza python tools/dump_track_streams.py plaza120 120000 154 0 1e5
7000 7001 7002 7003 on the MI355X, session s1), into a fixture of its own:
  python tools/make_track_box_fixture.py --out track_boxes_plaza.npz --first 2 gpurun_out/plaza120.npz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
argv = sys.argv[1:]
out_name, first = "track_boxes.npz", None
while argv and argv[0].startswith("--"):
    if argv[0] == "--out": out_name = argv[1]
    elif argv[0] == "--first": first = int(argv[1])
    else: raise SystemExit("unknown option " + argv[0])
    argv = argv[2:]
out, names = {}, []
for path in argv:
    d = np.load(path)
    F = len(d["ego_v"])
    kind = str(d["kind"]) if "kind" in d else "street"
    for b, scene in enumerate(d["scenes"]):
        if first is not None and b >= first:
            break
        name = f"{'scene' if kind == 'street' else kind}{int(scene)}_{int(d['points']) // 1000}k_unit{float(d['unit']):g}_preset{int(d['preset'])}"
        nb = [len(d[f"s{b}_f{f}_boxes_global"]) for f in range(F)]
        out[name + "/n_boxes"] = np.array(nb, np.int32)
        out[name + "/boxes_global"] = np.concatenate([d[f"s{b}_f{f}_boxes_global"] for f in range(F)]).astype(np.float32)
        out[name + "/meta"] = np.array([int(scene), int(d["points"]), float(d["unit"]), int(d["preset"]), F], np.float64)
        out[name + "/ego_v"] = d["ego_v"]; out[name + "/ego_yaw"] = d["ego_yaw"]
        names.append(name)
out["streams"] = np.array(names)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", out_name), **out)
print("tests/golden/" + out_name + ":", names)
