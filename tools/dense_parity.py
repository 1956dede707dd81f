"""The dense_scene leg's parity check alone (GPU box): python tools/dense_parity.py [scene id]  -> JSON with above_bar_detail
MOT_BENCH_LIB=variants/libmot_<name>.so: a variant build of the library (tools/prebuild.py), e.g. -DMOT_TRACK_SEQ_SUMS=1"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import numpy as np
import torch
mot = bench._load("mot_amd", os.path.join(bench.PKG_DIR, "__init__.py"))
sdev = bench._load("mot_amd.synth_dev", os.path.join(ROOT, "tools", "synth", "synth_dev.py"))
N, stride, F = 120000, 120832, 154
v, yaw = sdev.load_ego(F)
LIBP = os.environ.get("MOT_BENCH_LIB") or None
SCENE = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
seq, n_seq, _, _ = sdev.SequenceRenderer("cuda:0").render([SCENE], F, N, stride, v, yaw, scene="plaza")
n_seq = np.ascontiguousarray(n_seq, np.int32)
res = bench.gpu_sequence_results(mot, 0, seq, n_seq, stride, v, yaw, 0, lib_path=LIBP)
host = seq[:, 0, :N].cpu().numpy()
_b, par = bench.cpu_baseline(host, v, yaw, N, budget_s=4.0, gpu_results=res, n_per_frame=n_seq[:, 0], lib=mot.load_library(LIBP) if LIBP else mot.load_library(), quick=True)
print(json.dumps({"lib": LIBP or "product", "scene": SCENE, **{k: par.get(k) for k in ("measured","states_within_1e-4", "states_within_bar", "max_rel_state_err", "track_frames_above_1e-4", "track_frames_above_1e-4_not_set_aside",
                                             "track_frames_above_1e-4_unexplained", "set_aside_track_frames", "set_aside_by", "noise_floor_ill_conditioned", "state_compares", "above_bar_detail")}}, indent=1))
