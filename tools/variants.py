"""builds the library with extra -D flags per variant and times single kernels at B=128 (experiment tool)
usage: variants.py kernel_id[,kernel_id...] name=flag,flag ..."""
import importlib.util, os, sys, json
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
ids = [int(x) for x in sys.argv[1].split(",")]
for spec in sys.argv[2:]:
    name, _, fl = spec.partition("=")
    flags = [f for f in fl.split(",") if f]
    lib = build.build(extra_flags=flags, out=os.path.join(ROOT, "gpurun_out", f"libmot_v_{name}.so"))
    ctx = mot.Context(max_points=stride, max_batch=B, lib_path=lib)
    ctx.frames_dev(dev.data_ptr(), stride * 4, [N] * B); ctx.synchronize()
    print(name, {i: round(ctx.time_stage(i, B, 20), 4) for i in ids}, flush=True)
    ctx.close()
