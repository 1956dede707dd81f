"""GPU box: render bench / test streams (csrc/synth.hip), run them through the fused path frame by frame and save what the TRACKER saw
and produced — the boxes in the global frame (as track_prep_kernel handed them over), the track records and the full filter state of
every live track — to gpurun_out/<out>.npz. The tracker depends on nothing else, so the reference-side analysis (noise floors of the
reference's own arithmetic against the device's differences: tests/seq_parity.py NoiseFloor) can then run anywhere, repeatedly.
  python tools/dump_track_streams.py OUT POINTS FRAMES PRESET UNIT SCENE [SCENE …]      (MOT_DUMP_SCENE_KIND=plaza: the tracker-load scene)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
out, N, F, preset, unit = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
scenes = [int(x) for x in sys.argv[6:]]
import torch
from conftest import load_pkg, load_sub
import seq_parity as SP
mot = load_pkg(); sdev = load_sub("synth_dev")
stride = ((N + 2047) // 2048) * 2048
ego_v, ego_yaw = sdev.load_ego(F)
seq, n_seq, _o, _p = sdev.SequenceRenderer("cuda:0").render(scenes, F, N, stride, ego_v, ego_yaw, scene=os.environ.get("MOT_DUMP_SCENE_KIND", "street"))
n_seq = np.ascontiguousarray(n_seq, np.int32)
S = len(scenes)
save = {"scenes": np.array(scenes), "ego_v": ego_v, "ego_yaw": ego_yaw, "unit": unit, "preset": preset, "points": N, "kind": os.environ.get("MOT_DUMP_SCENE_KIND", "street")}
with mot.Context(mot.params(preset), max_points=stride, max_batch=S, max_tracks_total=1024) as c:
    for f in range(F):
        ts = np.full(S, 1.0e9 + f * unit)
        c.frames_dev(seq[f].data_ptr(), stride * 4, n_seq[f], run_tracker=True, timestamps=ts, ego_v=np.full(S, ego_v[f]), ego_yaw=np.full(S, ego_yaw[f]))
        for b in range(S):
            nb = len(c.get_boxes(b)["boxes"])
            gb = np.zeros((1024, 8, 3), np.float32)
            assert c.lib.mot_debug_copy(c._h, 11, b, gb.ctypes.data_as(C.c_void_p), C.c_size_t(gb.nbytes)) == 0
            tr = c.get_tracks(b)
            k = f"s{b}_f{f}_"
            save[k + "boxes_global"] = gb[:nb].copy()
            for q in ("track_manage", "is_static", "is_vis", "lifetime", "p", "v_yaw", "vis_box"):
                save[k + q] = tr[q]
            live = np.nonzero(tr["track_manage"] > 0)[0]
            save[k + "live"] = live
            for key in SP.STATE_KEYS:
                save[k + "st_" + key] = np.array([np.asarray(c.track_state(int(i), slot=b)[key], np.float64).reshape(-1) for i in live]).reshape(len(live), -1)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", out + ".npz"), **save)
print("saved", out, "streams", S, "frames", F)
