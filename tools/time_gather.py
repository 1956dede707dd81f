"""per-cluster phase timing of cluster_gather_kernel (experiment build with -DMOT_DBG_TIMING; results are wrong on purpose)"""
import importlib.util, os, sys, json, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
import torch
torch.cuda.init()
PKG = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd")
mot = _load("mot_amd", os.path.join(PKG, "__init__.py")); synth = _load("mot_amd.synth", os.path.join(PKG, "synth.py")); build = _load("mot_amd.build", os.path.join(PKG, "build.py"))
B, N = 64, 120000
stride = ((N + 2047) // 2048) * 2048
host = np.zeros((B, stride, 4), np.float32)
base = [synth.make_cloud(N, s, 0) for s in range(8)]
for b in range(B): host[b, :N] = base[b % 8]
dev = torch.from_numpy(host).cuda(); torch.cuda.synchronize()
lib = build.build(extra_flags=["-DMOT_DBG_TIMING"], out=os.path.join(ROOT, "gpurun_out", "libmot_timing.so"))
ctx = mot.Context(max_points=stride, max_batch=B, lib_path=lib)
ctx.frames_dev(dev.data_ptr(), stride * 4, [N] * B); ctx.synchronize()
# run label+gather only, then read candidates
import ctypes
l = ctx.lib
ms = ctx.time_stage(31, B, 3)   # leaves cand from gather overwritten by finalize? sequence: pre B1, timed B2, post B3 (B3 does not clear cand)
cand_dt = np.dtype([("pc", "f4", 8), ("max_z", "f4"), ("accepted", "i4"), ("undefined", "i4"), ("branch", "i4"), ("poly_off", "i4"), ("poly_n", "i4"), ("off_x", "i4"), ("off_y", "i4"), ("num_points", "i4"), ("pad", "i4")])
for slot in (0, 7):
    buf = np.zeros(32, cand_dt)
    rc = l.mot_debug_copy(ctx._h, 0, slot, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
    ncl = ctx.get_clusters(slot)["num_cluster"]
    for i in range(ncl):
        c = buf[i]
        if c["branch"] == 0: print(slot, i, "L  n=%d  t_prologue=%d t_rng=%d t_tiles=%d t_end=%d" % (c["num_points"], c["poly_off"], c["poly_n"], c["off_x"], c["off_y"]))
        else: print(slot, i, "MAR n=%d  t_prologue=%d t_tiles=%d t_compact=%d  wait=%d process=%d" % (c["num_points"], c["pad"], c["poly_off"], c["poly_n"], c["off_x"], c["off_y"]))
print("gather ms", ms)
