"""per-cluster phase clocks of cluster_rect_kernel (experiment build with -DMOT_DBG_RECT_TIMING: stamps in the cluster's run of the polygon pool)"""
import importlib.util, os, sys, ctypes
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
seq, n_seq, _, _ = sdev.SequenceRenderer("cuda").render(list(range(B)), 2, N, stride, v, yaw)
_pre = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "variants", "dbg_libmot_rect.so")
lib = _pre if os.path.exists(_pre) else build.build(extra_flags=["-DMOT_DBG_RECT_TIMING"], out=os.path.join(ROOT, "gpurun_out", "libmot_rect.so"))
ctx = mot.Context(max_points=stride, max_batch=B, lib_path=lib)
ctx.frames_dev(seq[1].data_ptr(), stride * 4, n_seq[1]); ctx.synchronize()
print("rect kernels ms", ctx.time_stage(33, B, 3))
cand_dt = np.dtype([("pc", "f4", 8), ("max_z", "f4"), ("accepted", "i4"), ("undefined", "i4"), ("branch", "i4"), ("poly_off", "i4"), ("poly_n", "i4"), ("off_x", "i4"), ("off_y", "i4"), ("num_points", "i4"), ("pad", "i4")])
rows = []
for slot in range(0, B, 4):
    cand = np.zeros(64, cand_dt); ctx.lib.mot_debug_copy(ctx._h, 0, slot, cand.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cand.nbytes))
    pool = np.zeros(stride, np.int32); ctx.lib.mot_debug_copy(ctx._h, 3, slot, pool.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(pool.nbytes))
    for i in range(min(ctx.get_clusters(slot)["num_cluster"], 64)):
        c = cand[i]
        if c["branch"] == 1 and c["poly_n"] >= 8:
            d = pool[c["poly_off"]: c["poly_off"] + 8]
            if d[0] == 0x7ec7: rows.append(d[1:8])
r = np.array(rows, np.float64)
names = ["loaded", "hull", "caliper setup", "caliper walk", "end"]
print(len(r), "rectangle clusters; candidates mean %.0f max %.0f; hull vertices mean %.0f max %.0f" % (r[:, 6].mean(), r[:, 6].max(), r[:, 5].mean(), r[:, 5].max()))
print("cycle stamps mean:", {n: int(r[:, k].mean()) for k, n in enumerate(names)}, " max:", {n: int(r[:, k].max()) for k, n in enumerate(names)})
big = r[np.argsort(-r[:, 4])[:5]]
print("five slowest [loaded, hull, setup, walk, end, hull vertices, candidates]:", [tuple(int(x) for x in row) for row in big])
