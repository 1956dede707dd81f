"""Offline (CPU): replay a dump of tools/dump_track_streams.py — the boxes the device's tracker saw and the states it produced — through
the reference's own tracker (oracle/_ref/libmot_ref.so) and its independent-arithmetic replicas (tests/seq_parity.py NoiseFloor), and
report, per stream: track-frames set aside by each criterion, the device's error on them against the reference's own noise floor there,
and every track-frame above 1e-4 with its explanation.   python tools/analyse_track_streams.py DUMP.npz [narrow|wide] [taint]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O, seq_parity as SP

def analyse(path, criterion="narrow", use_taint=False, stream=None, verbose=False):
    d = np.load(path)
    S = len(d["scenes"]); F = len(d["ego_v"]); unit = float(d["unit"]); preset = int(d["preset"])
    p = O.params(preset)
    out = []
    for b in range(S) if stream is None else [stream]:
        primary_ref = preset == 0 and O.ref() is not None
        T = O.RefTracker() if primary_ref else O.Tracker(p)
        if primary_ref: T.reset()
        NF = SP.NoiseFloor(O, p, primary_is_ref=primary_ref)
        stats, taint = {}, {}
        disc = None
        for f in range(F):
            k = f"s{b}_f{f}_"
            ts = 1.0e9 + f * unit
            T.ego_update(ts, float(d["ego_v"][f]), float(d["ego_yaw"][f]))
            gb = d[k + "boxes_global"]
            o = T.step(gb, ts, max_tracks=65536)
            NF.step(gb, ts, float(d["ego_v"][f]), float(d["ego_yaw"][f]), o, f)
            a = {q: d[k + q] for q in ("track_manage", "is_static", "is_vis", "lifetime", "p", "v_yaw", "vis_box")}; a["n"] = len(a["track_manage"])
            if "lifetime" in o: pass
            live = list(d[k + "live"])
            def sdev(i):
                j = live.index(i)
                st = {key: d[k + "st_" + key][j] for key in SP.STATE_KEYS}
                st["lifetime"] = int(a["lifetime"][i]); st["track_manage"] = int(a["track_manage"][i])
                return st
            try:
                SP.compare_tracks(a, o, sdev, T.state, (f, b), rtol=float("inf"), stats=stats, taint=taint if use_taint else None, frame=f if use_taint else None,
                                  criterion=criterion, floor=NF.floor)
            except AssertionError as e:
                disc = (f, str(e)[:160]); break
        r = {"stream": b, "scene": int(d["scenes"][b]), "criterion": criterion, "taint": use_taint, "primary": "reference build" if primary_ref else "restatement",
             "replicas": NF.names(), "replicas_retired_at_frame": NF.retired, "discrete_mismatch": disc,
             "live_track_frames": stats.get("state_compares", 0), "set_aside": stats.get("ill_conditioned", 0), "set_aside_by": stats.get("set_aside_by", {}),
             "max_rel_state_err": stats.get("max_rel_state_err"), "max_rel_state_err_set_aside": stats.get("max_rel_state_err_ill_conditioned"),
             "above_1e-4": stats.get("above_bar", 0), "above_1e-4_not_set_aside": stats.get("above_bar_well_conditioned", 0)}
        r.update(SP.floor_summary(stats))
        out.append(r)
        NF.close()
        if hasattr(T, "close"): T.close()
    return out

if __name__ == "__main__":
    crit = sys.argv[2] if len(sys.argv) > 2 else "narrow"
    for r in analyse(sys.argv[1], crit, len(sys.argv) > 3):
        print(json.dumps(r))
