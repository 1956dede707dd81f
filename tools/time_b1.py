"""per-workgroup phase timing of label_stats_kernel (experiment build with -DMOT_DBG_B1_TIMING)"""
import importlib.util, os, sys, json, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
import torch
torch.cuda.init()
PKG = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd")
mot = _load("mot_amd", os.path.join(PKG, "__init__.py")); synth = _load("mot_amd.synth", os.path.join(os.path.dirname(PKG), "tools", "synth", "synth.py")); build = _load("mot_amd.build", os.path.join(PKG, "build.py"))
B, N = 128, 120000
stride = ((N + 2047) // 2048) * 2048
host = np.zeros((B, stride, 4), np.float32)
base = [synth.make_cloud(N, s, 0) for s in range(8)]
for b in range(B): host[b, :N] = base[b % 8]
dev = torch.from_numpy(host).cuda(); torch.cuda.synchronize()
_pre = os.path.join(ROOT, "variants", "dbg_libmot_b1t.so")   # prebuilt on the CPU box (tools/prebuild_dbg.py): no hipcc minutes on the GPU box
lib = _pre if os.path.exists(_pre) else build.build(extra_flags=["-DMOT_DBG_B1_TIMING", "-DMOT_DBG_B1B_TIMING"], out=os.path.join(ROOT, "gpurun_out", "libmot_b1t.so"))
ctx = mot.Context(max_points=stride, max_batch=B, lib_path=lib)
ctx.frames_dev(dev.data_ptr(), stride * 4, [N] * B); ctx.synchronize()
ms = ctx.time_stage(30, B, 3)
print("label_stats ms", ms)
if len(sys.argv) > 1:
    print("index ms", ctx.time_stage(34, B, 3))
    for slot in (0, 5, 77):
        buf = np.zeros(8, np.int32)
        ctx.lib.mot_debug_copy(ctx._h, 3, slot, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
        print(slot, "index kernel cycles [scan, tables, prefixes, scatter, E, nwg, fast]:", buf[:7])
    sys.exit(0)
for slot in (0, 5, 77):
    ne = ctx.get_ground(slot, want_clouds=False)["n_elevated"]
    nwg = (ne + 2047) // 2048
    buf = np.zeros(nwg * 8, np.int32)
    ctx.lib.mot_debug_copy(ctx._h, 3, slot, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
    print(slot, "n_elev", ne, "cycles [first pt ready, loop end, barrier, aggregated, barrier, groups out, ng]:")
    print(buf.reshape(nwg, 8)[:, :7])
