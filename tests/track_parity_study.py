"""Where do the tracker's above-1e-4 track-frames on a dense scene come from?  (round-5 review, item 1c.)  CPU only.

  python tests/track_parity_study.py boxes [plaza|street] [scene id] [frames]    render the scene with tools/synth/synth_cpu.py (numpy mirror of the GPU
                                                                                   ray caster), run the REFERENCE's own ground removal -> clustering -> box fit
                                                                                   (oracle/_ref) on every frame -> gpurun_out/study_boxes_<scene>_<id>.npz
  python tests/track_parity_study.py run FIXTURE [stream]                        replay a box stream (the file above, or a stream of tests/golden/track_boxes.npz)
                                                                                   through: the reference build (primary), its replicas (C restatement, -DEIGEN_DONT_VECTORIZE
                                                                                   rebuild: the measured conditioning), and the DEVICE code on the emulator in four builds —
                                                                                   default (tree sums, host libm), -DMOT_TRACK_SEQ_SUMS=1, libm perturbed by an ulp, both.
Per build: track-frames above 1e-4 in total, on the measured-well-conditioned complement (replicas within 1e-5 of the primary there), and the worst
error / floor ratio. Test infrastructure: imports the oracle."""
import ctypes as C, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (lives in tests/ because it calls the oracle: test infrastructure, like everything that does)
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu"), os.path.join(ROOT, "tools", "synth")]
import oracle_lib as O
import seq_parity as SP
from conftest import load_pkg


def make_boxes(kind, scene_id, F, order="beam"):
    import synth_cpu
    frames, v, yaw = synth_cpu.render_sequence(scene_id, F, 120000, scene=kind, order=order)
    mot = load_pkg()
    import build_emu
    lib = C.CDLL(build_emu.build())   # (only for mot_debug_tf_matrix when oracle/_ref has no tf entry point; never steps anything here)
    R = O.RefTracker(); R.reset()
    out = {"ego_v": v, "ego_yaw": yaw, "kind": kind, "scene": scene_id}
    nb = []
    for f, c in enumerate(frames):
        g = O.ref_ground_remove(c); cl = O.ref_cluster(g["elevated"]); bx = O.ref_box_fit(g["elevated"], cl["grid"], cl["num_cluster"])["boxes"]
        ts = 1.0e9 + f * 1e5
        ego = R.ego_update(ts, float(v[f]), float(yaw[f]))
        gb = SP.boxes_to_global(O, lib, bx, ego[:3])
        out[f"f{f}"] = gb; nb.append(len(gb))
    out["n_boxes"] = np.array(nb, np.int32)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", f"study_boxes_{kind}_{scene_id}.npz")
    np.savez_compressed(path, **out)
    print("saved", path, "boxes per frame: mean %.1f max %d" % (np.mean(nb), max(nb)))


def load_stream(path, stream=None):
    d = np.load(path)
    if "streams" in d:   # tests/golden/track_boxes.npz
        name = str(d["streams"][int(stream or 0)]) if not isinstance(stream, str) or stream.isdigit() else stream
        scene, points, unit, preset, F = d[name + "/meta"]
        off = np.concatenate([[0], np.cumsum(d[name + "/n_boxes"])]); bx = d[name + "/boxes_global"]
        return name, float(unit), [bx[off[f]:off[f + 1]] for f in range(int(F))], d[name + "/ego_v"], d[name + "/ego_yaw"]
    if "scenes" in d:    # tools/dump_track_streams.py (GPU box): what the device's tracker was fed on rendered streams
        b = int(stream or 0); F = len(d["ego_v"])
        return f"{os.path.basename(path)}:scene{int(d['scenes'][b])}", float(d["unit"]), [d[f"s{b}_f{f}_boxes_global"] for f in range(F)], d["ego_v"], d["ego_yaw"]
    F = len(d["n_boxes"])
    return os.path.basename(path), 1e5, [d[f"f{f}"] for f in range(F)], d["ego_v"], d["ego_yaw"]


def emu_lib(defines="", perturb=False):
    import build_emu
    os.environ["MOT_EMU_DEFINES"] = defines
    if perturb:
        os.environ["MOT_EMU_PERTURB"] = "1"
    try:
        importlib.reload(build_emu)
        return build_emu.build()
    finally:
        os.environ.pop("MOT_EMU_DEFINES", None); os.environ.pop("MOT_EMU_PERTURB", None)
        importlib.reload(build_emu)


def run(path, stream=None):
    mot = load_pkg()
    name, unit, boxes, ego_v, ego_yaw = load_stream(path, stream)
    p = O.params(0)
    builds = {"default (tree sums)": emu_lib(), "seq sums": emu_lib("-DMOT_TRACK_SEQ_SUMS=1"), "default + libm 1 ulp off": emu_lib(perturb=True),
              "seq sums + libm 1 ulp off": emu_lib("-DMOT_TRACK_SEQ_SUMS=1", perturb=True)}
    report = {"stream": name, "frames": len(boxes)}
    for tag, lib in builds.items():
        R = O.RefTracker(); R.reset()
        NF = SP.NoiseFloor(O, p, primary_is_ref=True)
        st = dict(live=0, above=0, above_measured_well=0, ill_measured=0, worst_ratio=0.0, worst_well=0.0, worst=0.0, set_aside_narrow=0, above_not_narrow=0, discrete_equal=True)
        with mot.Context(lib_path=lib, max_points=1024, max_tracks_total=1024) as c:
            for f, gb in enumerate(boxes):
                ts = 1.0e9 + f * unit
                R.ego_update(ts, float(ego_v[f]), float(ego_yaw[f])); c.ego_update(ts, float(ego_v[f]), float(ego_yaw[f]))
                o = R.step(gb, ts, max_tracks=65536); a = c.track_step(gb, ts)
                NF.step(gb, ts, float(ego_v[f]), float(ego_yaw[f]), o, f)
                if a["n"] != o["n"] or any(not np.array_equal(a[q], o[q]) for q in ("track_manage", "is_static", "is_vis")):
                    st["discrete_equal"] = False; st["parted_at_frame"] = f
                    break
                for i in np.nonzero(o["track_manage"] > 0)[0]:
                    so = R.state(int(i)); sd = c.track_state(int(i))
                    e, same = SP.state_rel_err(sd, so)
                    fl = NF.floor(int(i), so)
                    st["live"] += 1
                    narrow = bool(SP.set_aside_reasons(so, "narrow"))
                    st["set_aside_narrow"] += narrow
                    ill = fl is None or not np.isfinite(fl) or fl > SP.MEASURED_FLOOR
                    st["ill_measured"] += ill
                    if not same:
                        e = float("inf")
                    st["worst"] = max(st["worst"], e if np.isfinite(e) else 1e9)
                    if e > SP.RTOL:
                        st["above"] += 1
                        st["above_not_narrow"] += not narrow
                        if not ill:
                            st["above_measured_well"] += 1
                        if fl:
                            st["worst_ratio"] = max(st["worst_ratio"], e / fl)
                    if not ill:
                        st["worst_well"] = max(st["worst_well"], e)
        st["replicas_retired"] = dict(NF.retired)
        NF.close()
        report[tag] = st
        print(tag, json.dumps(st), flush=True)
    return report


if __name__ == "__main__":
    if sys.argv[1] == "boxes":
        make_boxes(sys.argv[2] if len(sys.argv) > 2 else "plaza", int(sys.argv[3]) if len(sys.argv) > 3 else 7000, int(sys.argv[4]) if len(sys.argv) > 4 else 154)
    else:
        r = run(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
        print(json.dumps(r))
