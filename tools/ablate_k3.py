"""timing ablations of classify_compact_kernel (experiment builds; some give wrong results on purpose)"""
import importlib.util, os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
import torch
torch.cuda.init()
PKG = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd")
mot = _load("mot_amd", os.path.join(PKG, "__init__.py")); synth = _load("mot_amd.synth", os.path.join(PKG, "synth.py")); build = _load("mot_amd.build", os.path.join(PKG, "build.py"))
B, N = 128, 120000
stride = ((N + 2047) // 2048) * 2048
host = np.zeros((B, stride, 4), np.float32)
base = [synth.make_cloud(N, s, 0) for s in range(8)]
for b in range(B): host[b, :N] = base[b % 8]
dev = torch.from_numpy(host).cuda(); torch.cuda.synchronize()
out = {}
# plain device copy of the same byte volume (read 16 B/pt, write 16 B/pt) as the practical ceiling for mixed traffic
dst = torch.empty_like(dev)
for _ in range(3): dst.copy_(dev)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): dst.copy_(dev)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
out["torch_copy"] = {"ms": ms, "GBps": 2 * dev.numel() * 4 / ms / 1e6}
s = dev.sum()
e0.record()
for _ in range(20): s = dev.sum()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
out["torch_sum_read"] = {"ms": ms, "GBps": dev.numel() * 4 / ms / 1e6}
del dst
variants = [("full", []), ("nolookback", ["-DMOT_DBG_K3_NOLOOKBACK"]), ("cheapcell", ["-DMOT_DBG_CHEAPCELL"]),
            ("both", ["-DMOT_DBG_K3_NOLOOKBACK", "-DMOT_DBG_CHEAPCELL"])]
for name, flags in variants:
    lib = build.build(extra_flags=flags, out=os.path.join(ROOT, "gpurun_out", f"libmot_{name}.so"))
    ctx = mot.Context(max_points=stride, max_batch=B, lib_path=lib)
    ctx.frames_dev(dev.data_ptr(), stride * 4, [N] * B); ctx.synchronize()
    out[name] = {"k1_ms": ctx.time_stage(10, B, 20), "k3_ms": ctx.time_stage(12, B, 20)}
    ctx.close()
print(json.dumps(out))
