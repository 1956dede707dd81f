"""run by tests/test_sequence_gpu.py::test_sequence_mode_on_the_device in a process of its own (torch for the renderer)"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, HERE)
import torch
from conftest import load_pkg, load_sub
import seq_parity as SP
mot = load_pkg(); sdev = load_sub("synth_dev")
F, N = 154, 120000
stride = ((N + 2047) // 2048) * 2048
ego_v, ego_yaw = sdev.load_ego(F)
seq, n_seq, _o, _p = sdev.SequenceRenderer("cuda:0").render([5], F, N, stride, ego_v, ego_yaw)
n0 = np.ascontiguousarray(n_seq[:, 0], np.int32); ts = 1.0e9 + 1e5 * np.arange(F)
per = []
with mot.Context(max_points=stride, max_batch=1, max_tracks_total=256) as c:
    for f in range(F):
        c.frames_dev(seq[f, 0].data_ptr(), stride * 4, [n0[f]], run_tracker=True, timestamps=[ts[f]], ego_v=[ego_v[f]], ego_yaw=[ego_yaw[f]])
        tr = c.get_tracks(0)
        per.append((c.get_boxes(0)["boxes"], tr))
    states = {int(i): c.track_state(int(i)) for i in np.nonzero(tr["track_manage"] > 0)[0]}
rec = np.dtype([("id", "i4"), ("track_manage", "i4"), ("is_static", "i4"), ("is_vis", "i4"), ("p", "f4", 3), ("lifetime", "i4"), ("v_yaw", "f8", 2), ("vis_box", "f4", 24)])
with mot.Context(max_points=stride, max_batch=F, max_tracks_total=256) as c:
    out = torch.zeros(F * 64 * rec.itemsize, dtype=torch.uint8, device="cuda"); cnt = torch.full((F,), -1, dtype=torch.int32, device="cuda")
    c.sequence_dev(seq[0, 0].data_ptr(), int(seq.stride(0)), n0, ts, ego_v, ego_yaw, out.data_ptr(), 64, cnt.data_ptr())
    c.synchronize()
    o = out.cpu().numpy().view(rec).reshape(F, 64); k = cnt.cpu().numpy()
    for f in range(F):
        bx, tr = per[f]
        assert SP.bits_equal(c.get_boxes(f)["boxes"], bx), (f, "boxes")
        live = np.nonzero(tr["track_manage"] > 0)[0]
        assert k[f] == len(live) and np.array_equal(o[f]["id"][: k[f]], live), (f, k[f], len(live))
        bits = lambda a: np.ascontiguousarray(a).view(np.uint8)   # (a diverging track's NaN outputs must be the same NaNs)
        for key in ("track_manage", "is_static", "is_vis", "lifetime", "p", "v_yaw", "vis_box"):
            assert np.array_equal(bits(o[f][key][: k[f]]), bits(tr[key][live])), (f, key)
    for i, so in states.items():
        sd = c.track_state(i)
        for key in SP.STATE_KEYS:
            assert np.array_equal(bits(np.asarray(sd[key])), bits(np.asarray(so[key]))), (i, key)
print("sequence mode ok:", F, "frames,", int(tr["n"]), "tracks ever,", len(states), "live at the end")
