"""How many live tracks does a rendered scene show the tracker? (GPU box)   python tools/scene_tracks.py [street|plaza] [streams] [frames]"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m)
    return m


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "plaza"
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    F = int(sys.argv[3]) if len(sys.argv) > 3 else 154
    import torch
    mot = _load("mot_amd", os.path.join(PKG, "__init__.py"))
    sdev = _load("mot_amd.synth_dev", os.path.join(ROOT, "tools", "synth", "synth_dev.py"))
    N, stride = 120000, 120832
    v, yaw = sdev.load_ego(F)
    seq, n, objs, path = sdev.SequenceRenderer("cuda:0").render(list(range(S)), F, N, stride, v, yaw, scene=scene)
    print(scene, "objects per scene:", [int((o[:, 5] > 0).sum()) for o in objs])
    with mot.Context(device=0, max_points=stride, max_batch=S, max_tracks_total=256) as c:
        rows = []
        for f in range(F):
            ts = np.full(S, 1.0e9 + f * 1e5)
            c.frames_dev(seq[f].data_ptr(), stride * 4, n[f], run_tracker=True, timestamps=ts, ego_v=np.full(S, v[f]), ego_yaw=np.full(S, yaw[f]))
            if f % 10 == 0 or f == F - 1:
                live, boxes, ne = [], [], []
                for b in range(S):
                    tr = c.get_tracks(b)
                    live.append(int((tr["track_manage"] > 0).sum())); boxes.append(len(c.get_boxes(b)["boxes"]))
                    ne.append(c.get_ground(b, int(n[f, b]), want_clouds=False)["n_elevated"])
                rows.append((f, np.mean(live), max(live), np.mean(boxes), np.mean(ne), int(tr["n"])))
                print("frame %3d  live tracks mean %.1f max %d  boxes mean %.1f  elevated mean %.0f  tracks ever (last stream) %d" % rows[-1], flush=True)


if __name__ == "__main__":
    main()
