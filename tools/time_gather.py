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
mot = _load("mot_amd", os.path.join(PKG, "__init__.py")); sdev = _load("mot_amd.synth_dev", os.path.join(os.path.dirname(PKG), "tools", "synth", "synth_dev.py")); build = _load("mot_amd.build", os.path.join(PKG, "build.py"))
B, N = 128, 120000
stride = ((N + 2047) // 2048) * 2048
v, yaw = sdev.load_ego(2)
seq, n_seq, _, _ = sdev.SequenceRenderer("cuda").render(list(range(B)), 2, N, stride, v, yaw)   # frames of the bench workload
_pre = os.path.join(ROOT, "variants", "dbg_libmot_timing.so")   # prebuilt on the CPU box (tools/prebuild_dbg.py): no hipcc minutes on the GPU box
lib = _pre if os.path.exists(_pre) else build.build(extra_flags=["-DMOT_DBG_TIMING"], out=os.path.join(ROOT, "gpurun_out", "libmot_timing.so"))
ctx = mot.Context(max_points=stride, max_batch=B, lib_path=lib)
ctx.frames_dev(seq[1].data_ptr(), stride * 4, n_seq[1]); ctx.synchronize()
# run label+gather only, then read candidates
import ctypes
l = ctx.lib
ms = ctx.time_stage(31, B, 3)   # leaves cand from gather overwritten by finalize? sequence: pre B1, timed B2, post B3 (B3 does not clear cand)
cand_dt = np.dtype([("pc", "f4", 8), ("max_z", "f4"), ("accepted", "i4"), ("undefined", "i4"), ("branch", "i4"), ("poly_off", "i4"), ("poly_n", "i4"), ("off_x", "i4"), ("off_y", "i4"), ("num_points", "i4"), ("pad", "i4")])
rows = []
for slot in range(0, B, 4):
    buf = np.zeros(64, cand_dt)
    rc = l.mot_debug_copy(ctx._h, 0, slot, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
    ncl = min(ctx.get_clusters(slot)["num_cluster"], 64)
    for i in range(ncl):
        c = buf[i]
        if c["branch"] == 0: rows.append(("L", int(c["num_points"]), int(c["poly_off"]), int(c["poly_n"]), int(c["off_x"]), int(c["off_y"])))
        elif c["branch"] == 1: rows.append(("MAR", int(c["num_points"]), int(c["pad"]), int(c["poly_off"]), int(c["poly_n"]), 0))
for kind in ("L", "MAR"):
    r = np.array([x[1:] for x in rows if x[0] == kind], np.float64)
    if len(r) == 0: continue
    print(kind, "clusters", len(r), "points mean %.0f max %.0f" % (r[:, 0].mean(), r[:, 0].max()))
    names = ["prologue", "rng", "sample lookups", "end"] if kind == "L" else ["prologue", "point walk", "compaction", "-"]
    print("   cycle stamps (mean / max):", {n: (int(r[:, 1 + k].mean()), int(r[:, 1 + k].max())) for k, n in enumerate(names)})
    big = r[np.argsort(-r[:, 0])[:5]]
    print("   five largest:", [tuple(int(v) for v in row) for row in big])
print("gather ms", ms, "clusters per frame", len(rows) / (B / 4))
