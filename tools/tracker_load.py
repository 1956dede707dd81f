"""tracker step (the four launches of mot_launch_track, HIP events around them) under load:
 (A) mot_track_steps_dev on T = 8 / 32 / 64 slowly moving boxes per stream, (B) a rendered 154-frame bench sequence.
   python tools/tracker_load.py [streams] [extra hipcc flags for a variant build ...]"""
import ctypes, importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
import torch
torch.cuda.init()
PKG = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd")
mot = _load("mot_amd", os.path.join(PKG, "__init__.py")); sdev = _load("mot_amd.synth_dev", os.path.join(os.path.dirname(PKG), "tools", "synth", "synth_dev.py"))
build = _load("mot_amd.build", os.path.join(PKG, "build.py"))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
flags = [a for a in sys.argv[2:] if a.startswith("-D")]
STRESS_ONLY = "stress-only" in sys.argv
lib = next((a[4:] for a in sys.argv[2:] if a.startswith("lib=")), None)   # a prebuilt variant library (variants/, built on the CPU box)
if flags:
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    lib = build.build(extra_flags=flags, out=os.path.join(ROOT, "gpurun_out", "libmot_variant_" + "_".join(f.strip("-D").replace("=", "") for f in flags) + ".so"))
print("variant:", flags or lib or "product build")
kw = dict(lib_path=lib) if lib else {}

rng = np.random.default_rng(11)
for T in (8, 32, 64):
    with mot.Context(max_points=1024, max_batch=S, max_tracks_total=1024, **kw) as c:
        side = int(np.ceil(np.sqrt(T)))
        centres = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2)[:T] * 9.0 - side * 4.5
        vel = rng.uniform(-1.0, 1.0, size=(S, T, 2))
        for f in range(40):
            ts = 1.0e9 + f * 1e5
            ctr = centres[None] + vel * (0.1 * f) + rng.normal(0, 0.02, size=(S, T, 2))
            bx = np.zeros((S, T, 8, 3), np.float32)
            bx[..., :2] = ctr[:, :, None, :] + np.array([[0, 0], [3.8, 0], [3.8, 1.7], [0, 1.7]] * 2)[None, None]
            bx[:, :, :4, 2] = -2.0; bx[:, :, 4:, 2] = -0.4
            d = torch.from_numpy(bx.reshape(S, T * 24)).cuda()
            for s in range(S): c.ego_update(ts, 0.0, 0.0, s)
            if f == 14: c.synchronize(); c.profile_kernel(40, 1)
            c.track_steps_dev(d.data_ptr(), T * 24, [T] * S, [ts] * S)
            c.synchronize()
        r = c.profile_read(); tr = c.get_tracks(0)
        print(f"stress T={T}: {r['mean_ms']*1e3:.1f} us mean, {r['min_ms']*1e3:.1f} min, {r['max_ms']*1e3:.1f} max over {r['samples']} steps of {S} streams; "
              f"live {int((tr['track_manage'] > 0).sum())}, ever {tr['n']}")

if STRESS_ONLY:
    sys.exit(0)
B, N, F = 128 if S >= 128 else S, 120000, 154
stride = ((N + 2047) // 2048) * 2048
v, yaw = sdev.load_ego(F)
seq, n_seq, _, _ = sdev.SequenceRenderer("cuda").render(list(range(B)), F, N, stride, v, yaw)
with mot.Context(max_points=stride, max_batch=B, max_tracks_total=4096, **kw) as c:
    c.profile_kernel(40, 3)
    for f in range(F):
        c.frames_dev(seq[f].data_ptr(), stride * 4, n_seq[f], run_tracker=True, timestamps=[1e9 + f * 1e5] * B, ego_v=[v[f]] * B, ego_yaw=[yaw[f]] * B)
        c.synchronize()
    r = c.profile_read()
    live = [int((c.get_tracks(b)["track_manage"] > 0).sum()) for b in range(0, B, 8)]
    print(f"sequence, {B} streams x {F} frames: tracker step {r['mean_ms']*1e3:.1f} us mean, {r['min_ms']*1e3:.1f} min, {r['max_ms']*1e3:.1f} max ({r['samples']} samples); "
          f"live at the end mean {np.mean(live):.1f} max {max(live)}, ever {c.get_tracks(0)['n']}")
