"""phase clocks of track_step_stream_kernel (the one-launch tracker step) on a rendered single stream — needs variants/libmot_strt.so
(tools/prebuild.py strt=-DMOT_DBG_STREAM_TIMING). Prints, per sampled frame, microseconds since the kernel's start at: prologue done,
prediction done (thread 0's wave), barrier passed, update done, barrier passed, finish done; and the live tracks."""
import ctypes as C, importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
import torch
PKG = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd")
mot = _load("mot_amd", os.path.join(PKG, "__init__.py")); sdev = _load("mot_amd.synth_dev", os.path.join(os.path.dirname(PKG), "tools", "synth", "synth_dev.py"))
lib = os.path.join(ROOT, "variants", "libmot_strt.so")
N, F = 120000, 154
stride = ((N + 2047) // 2048) * 2048
v, yaw = sdev.load_ego(F)
seq, n_seq, _, _ = sdev.SequenceRenderer("cuda").render([0], F, N, stride, v, yaw)
rows = []
with mot.Context(max_points=stride, max_batch=1, max_tracks_total=256, lib_path=lib) as c:
    for f in range(F):
        c.frames_dev(seq[f].data_ptr(), stride * 4, n_seq[f], run_tracker=True, timestamps=[1e9 + f * 1e5], ego_v=[v[f]], ego_yaw=[yaw[f]])
        d = np.zeros(32, np.int64)
        assert c.lib.mot_debug_copy(c._h, 12, 0, d.ctypes.data_as(C.c_void_p), C.c_size_t(256)) == 0
        rows.append(d.copy())
r = np.array(rows[10:], np.float64)
print("us since kernel start (100 MHz clock), mean over frames 10..153: prologue %.1f | predict done %.1f | barrier %.1f | update done %.1f | barrier %.1f | finish done %.1f | live %.1f"
      % tuple(list(r[:, :6].mean(0) / 100.0) + [r[:, 6].mean()]))
print("finish phase from its start: claimed boxes %.1f | visible boxes %.1f | merge pairs %.1f | apply %.1f | birth %.1f | outputs + live list %.1f" % tuple(r[:, 8:14].mean(0) / 100.0))
a = (r[:, 14:31] - r[:, 14:15]) / 100.0   # absolute clocks of thread 0 (wave 0, group 0) relative to the kernel's start
m = a.mean(0)
print("prediction (thread 0's group; us since it entered the prediction): enter %.1f | track loaded %.1f | det5 guard %.1f | mixing + interaction %.1f | Cholesky %.1f | sigma points %.1f | mean, S, Tc, K %.1f | covariance %.1f | models stored %.1f | gating done %.1f"
      % (m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9]))
print("update: enter %.1f | claimed-before bookkeeping %.1f | track loaded %.1f | association + updateBB %.1f | PDA sums %.1f | state / covariance update %.1f | mode probabilities + merge + stores %.1f" % (m[10], m[11], m[12], m[13], m[15], m[16], m[14]))
for f in (20, 60, 100, 150):
    print("frame", f, (rows[f][:6] / 100.0).round(1), "live", rows[f][6])
