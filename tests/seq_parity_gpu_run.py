"""Run by tests/test_sequence_gpu.py in a process of its own (it needs torch for the GPU renderer; the other GPU tests keep
torch out of their process, see tests/hiprt.py): BASELINE.json configs[3] AS WRITTEN on the MI355X — synth_dev-rendered
sequences (the generator bench.py times: csrc/synth.hip) of `--frames` frames with the ego motion of KITTI drive_0005 through
the fused device path, EVERY frame of every stream compared with the oracle by tests/seq_parity.py: mask, clouds, label grid,
per-point labels, boxes, boxes in the global frame bit-exact; track set, trackManage, lifetime, static / vis flags exact; every
state key <= 1e-4 relative.  TEST INFRASTRUCTURE (imports the oracle)."""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=120000)
    ap.add_argument("--frames", type=int, default=154)
    ap.add_argument("--scenes", type=int, nargs="+", default=[0, 1])
    ap.add_argument("--units", type=float, nargs="+", default=[1e5, 0.1], help="timestamp step per stream (SURVEY.md H11: 1e5 us or 0.1 s)")
    ap.add_argument("--density", type=float, default=1.0)
    ap.add_argument("--preset", type=int, default=0)
    ap.add_argument("--scene", choices=("street", "plaza"), default="street", help="plaza: the tracker-load scene (50-65 live tracks per stream: configs[3]'s '<= 64 tracks')")
    ap.add_argument("--order", choices=("beam", "firing", "random"), default="beam", help="point order in memory (tools/synth/synth_dev.py)")
    ap.add_argument("--measured", action="store_true", help="the MEASURED conditioning (tests/seq_parity.py MEASURED_FLOOR): every live track-frame on which the reference's own "
                    "builds agree to 1e-5 is held to 1e-4 strictly")
    args = ap.parse_args()
    import torch
    from conftest import load_pkg, load_sub
    import oracle_lib as O
    import seq_parity as SP
    mot = load_pkg(); sdev = load_sub("synth_dev"); build = load_sub("build")
    if not os.path.exists(build.LIB):
        raise SystemExit("libmot_hip.so is missing: no fallback")
    O.build_oracle()
    if O.ref() is not None and os.environ.get("MOT_ORACLE") != "restatement":
        O = O.RefFirst(O)   # the reference's own sources wherever they can answer (tests/oracle_lib.py); the restatement otherwise
    S, F, N = len(args.scenes), args.frames, args.points
    stride = ((N + 2047) // 2048) * 2048
    ego_v, ego_yaw = sdev.load_ego(F)
    t0 = time.time()
    seq, n_seq, _objs, _path = sdev.SequenceRenderer("cuda:0").render(args.scenes, F, N, stride, ego_v, ego_yaw, density=args.density, scene=args.scene, order=args.order)
    n_seq = np.ascontiguousarray(n_seq, np.int32)
    t_render = time.time() - t0
    units = (args.units * S)[:S]
    p = O.params(args.preset)
    t0 = time.time()
    with mot.Context(mot.params(args.preset), max_points=stride, max_batch=S, max_tracks_total=1024) as c:
        st = SP.check_sequence(c, O, p, lambda f: seq[f].data_ptr(), lambda f, b: seq[f, b].cpu().numpy(), n_seq, stride, ego_v, ego_yaw, units, skip_ill_conditioned=True, noise_floor=True, mar_check=True, measured=args.measured)
    st.update(render_s=round(t_render, 1), check_s=round(time.time() - t0, 1), points_per_frame=int(n_seq.mean()), scenes=args.scenes, units=units,
              reference_tf=O.ref_tf() is not None, scene=args.scene, point_order=args.order,
              oracle_used={f"{fn}: {who}": k for (fn, who), k in sorted(getattr(O, "used", {}).items())} or "restatement")
    if args.preset == 0:
        assert st["boxes"] > F and st["tracks_ever"] >= 20 and st["live_max"] >= 5, st   # the sequence really exercises the tracker
    print("sequence parity ok " + json.dumps(st))


if __name__ == "__main__":
    main()
