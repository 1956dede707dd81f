"""compaction kernel per chunk vs per frame (mot_debug_option 0), solo, at `B` frames per launch: kernel 12, the labelling kernel
21 (which folds lists / reads planes) and the ground stage, HIP events via mot_time_stage.   python tools/time_k3_modes.py [B] [lib]"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
import torch
torch.cuda.init()
PKG = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd")
mot = _load("mot_amd", os.path.join(PKG, "__init__.py")); sdev = _load("mot_amd.synth_dev", os.path.join(PKG, "synth_dev.py"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
libs = sys.argv[2:] or [None]
N, F = 120000, 2
stride = ((N + 2047) // 2048) * 2048
v, yaw = sdev.load_ego(F)
seq, n_seq, _, _ = sdev.SequenceRenderer("cuda").render(list(range(B)), F, N, stride, v, yaw)
for lib in [l or None for l in libs]:
    with mot.Context(max_points=stride, max_batch=B, **({"lib_path": lib} if lib else {})) as c:
        for mode, name in ((0, "chunk"), (1, "frame")):
            assert c.lib.mot_debug_option(c._h, 0, mode) == 0
            c.frames_dev(seq[1].data_ptr(), stride * 4, n_seq[1]); c.synchronize()
            t = {k: c.time_stage(k, B, 10) * 1e3 for k in (12, 21, 30, 34, 31, 0, 2, 100)}
            print(f"{os.path.basename(lib) if lib else 'product':24s} {name}: K3 {t[12]:6.1f}  ccl {t[21]:5.1f}  label {t[30]:6.1f}  index {t[34]:5.1f}  gather {t[31]:6.1f} | ground {t[0]:6.1f}  box {t[2]:6.1f}  stateless {t[100]:7.1f} us  (B={B})", flush=True)
