"""phase timing of track_step_kernel from its shader-clock stamps (diagnostics)"""
import importlib.util, os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
import torch
torch.cuda.init()
PKG = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd")
mot = _load("mot_amd", os.path.join(PKG, "__init__.py")); synth = _load("mot_amd.synth", os.path.join(PKG, "synth.py"))
B, N, F = 64, 120000, 4
stride = ((N + 2047) // 2048) * 2048
devs = []
for f in range(F):
    host = np.zeros((B, stride, 4), np.float32)
    for b in range(B): host[b, :N] = synth.make_cloud(N, b % 8, f)
    devs.append(torch.from_numpy(host).cuda())
torch.cuda.synchronize()
build = _load("mot_amd.build", os.path.join(PKG, "build.py"))
lib = build.build(extra_flags=[sys.argv[1] if len(sys.argv) > 1 else "-DMOT_DBG_TRACK_SUB"], out=os.path.join(ROOT, "gpurun_out", "libmot_tsub.so"))
ctx = mot.Context(max_points=stride, max_batch=B, max_tracks_total=8192, lib_path=lib)
for k in range(24):
    ts = [1e9 + k * 1e5] * B
    ctx.frames_dev(devs[k % F].data_ptr(), stride * 4, [N] * B, run_tracker=True, timestamps=ts, ego_v=[0.0] * B, ego_yaw=[0.0] * B)
ctx.synchronize()
names = ["P0 compact", "PA predict+gate", "PB bookkeeping", "PC update", "PD merge", "PE birth", "PF outputs"]
for slot in (0, 3, 5, 7):
    buf = np.zeros(16, np.int64)
    ctx.lib.mot_debug_copy(ctx._h, 2, slot, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
    d = np.diff(buf[:8])
    if len(sys.argv) > 1 and "UKF" in sys.argv[1]:
        print("   imm_ukf (last track of wave 0): mixing %d | cholesky %d | sigma points %d | mean+cov %d | update-lidar part .. PA end %d" % (buf[12] - buf[11], buf[13] - buf[12], buf[14] - buf[13], buf[15] - buf[14], buf[2] - buf[15]))
    else: print("   PC sub (last track of wave 0): load_track %d | gate words %d | association %d | update_bb %d | management %d | PDA+rest %d" % (buf[11] - buf[3], buf[12] - buf[11], buf[13] - buf[12], buf[14] - buf[13], buf[15] - buf[14], buf[4] - buf[15]))
    print("slot", slot, "nlive", buf[8], "ntracks", buf[9], "boxes", buf[10], {n: int(v) for n, v in zip(names, d)}, "total", int(buf[7] - buf[0]))
print("tracker kernel ms", ctx.time_stage(40, B, 3))
