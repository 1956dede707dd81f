"""Ablations on SCRATCH COPIES of the kernel sources (the product sources carry no ablation forks): every experiment is a
list of text substitutions applied to a copy of csrc/ under gpurun_out/, built into its own library and timed kernel by
kernel (mot_time_stage) on frames of the bench workload. Most of these variants give WRONG results on purpose.
   python tools/ablate.py kernel_id[,kernel_id...] [experiment ...]        (no experiment names = all)"""
import importlib.util, os, shutil, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
import torch
torch.cuda.init()
PKG = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd")
mot = _load("mot_amd", os.path.join(PKG, "__init__.py")); sdev = _load("mot_amd.synth_dev", os.path.join(os.path.dirname(PKG), "tools", "synth", "synth_dev.py")); build = _load("mot_amd.build", os.path.join(PKG, "build.py"))

EXPERIMENTS = {
    "baseline": [],
    # ---- geometry knobs (build flags, correct results)
    "k3_256x16": [("FLAGS", "-DMOT_COMPACT_BLOCK=256", "-DMOT_COMPACT_CHUNK=4096")],
    "k3_1024x8": [("FLAGS", "-DMOT_COMPACT_BLOCK=1024", "-DMOT_COMPACT_CHUNK=8192")],
    # ---- classify_compact_kernel (12)
    "k3_no_occupancy": [("ground.hip", "const bool occupancy = g.occ_list != nullptr;", "const bool occupancy = false;")],
    "k3_no_lookback": [("ground.hip", "    if (chunk > 0) {\n      if (lane == 0) __hip_atomic_store(&desc[chunk], kDescAggregate | mine",
                        "    excl_e = chunk * 1200; excl_g = chunk * 2896;\n    if (false) {\n      if (lane == 0) __hip_atomic_store(&desc[chunk], kDescAggregate | mine")],
    "k3_no_mask": [("ground.hip", "    if (mask) {\n      if (full) {", "    if (false) {\n      if (full) {")],
    "k3_no_ground_store": [("ground.hip", "    if (cls[k] != MOT_MASK_DROPPED) dst[at] = pt[k];", "    if (is_e) dst[at] = pt[k];")],
    "k3_no_stores": [("ground.hip", "    if (cls[k] != MOT_MASK_DROPPED) dst[at] = pt[k];", "    if (at == -12345) dst[at] = pt[k];")],
    "k3_plain_loads": [("ground.hip", "pt[k] = load_stream(&in[base + k * kCompactBlock + threadIdx.x]);  // last use of the input cloud", "pt[k] = in[base + k * kCompactBlock + threadIdx.x];")],
    "k3_cheap_cell": [("ground.hip", "    int c = mot_polar_cell_try(p, pt[k].x, pt[k].y);", "    int c = ((int)(pt[k].x * 0.5f) & 63) * MOT_NUM_BIN + ((int)(pt[k].y * 0.5f) & 63);")],
    # ---- LDS footprints of the low-occupancy kernels (do they keep other contexts' kernels off their CUs?)
    "rect_small_lds": [("box.hip", "  __shared__ int s_in[kMaxHullIn + 2];                        // candidate points (x | y << 16), sorted by (x,y)", "  __shared__ int s_in[512];"),
                       ("box.hip", "  __shared__ int s_b0[kMaxHullIn + 2], s_b1[kMaxHullIn + 2];  // peeling ping-pong", "  __shared__ int s_b0[512], s_b1[512];"),
                       ("box.hip", "    if (cand.poly_off + total > c.cap || total > kMaxHullIn) total = 0;", "    if (cand.poly_off + total > c.cap || total > 510) total = 0;")],
    "track_items_1wave": [("FLAGS", "-DMOT_TRACK_ITEM_WAVES=1", "-DMOT_PREDICT_WAVES=2")],
    # ---- label_stats_kernel (30)
    "b1_no_pix": [("box.hip", "      pix[i] = (int)((unsigned)((picX >= 0 && picX < 1024) ? picX : 0xffff) | ((unsigned)picY << 16));", "")],
}

BENCH = sys.argv[1] == "bench"   # `ablate.py bench name ...`: run bench.py (short) on each variant instead of timing single kernels
ids = [] if BENCH else [int(x) for x in sys.argv[1].split(",")]
names = sys.argv[2:] or list(EXPERIMENTS)
B, N, F = 128, 120000, 2
stride = ((N + 2047) // 2048) * 2048
v, yaw = sdev.load_ego(F)
if not BENCH:
    seq, n_seq, _, _ = sdev.SequenceRenderer("cuda").render(list(range(B)), F, N, stride, v, yaw)
for name in names:
    src = os.path.join(ROOT, "gpurun_out", "ablate_" + name)
    shutil.rmtree(src, ignore_errors=True)
    shutil.copytree(os.path.join(PKG, "csrc"), src, ignore=shutil.ignore_patterns(".*", "__pycache__"))
    # the scratch copy sits two levels deeper than csrc/: give it the headers it includes relatively
    os.makedirs(os.path.join(ROOT, "gpurun_out", "include"), exist_ok=True)
    flags = []
    for fn, old, new in EXPERIMENTS[name]:
        if fn == "FLAGS":
            flags += [old, new]
            continue
        p = os.path.join(src, fn); s = open(p).read()
        assert s.count(old) == 1, (name, fn, s.count(old))
        open(p, "w").write(s.replace(old, new))
    for fn in os.listdir(src):   # "../../include/mot.h" relative to csrc/
        p = os.path.join(src, fn)
        if not os.path.isfile(p):
            continue
        s = open(p).read()
        if '"../../include/mot.h"' in s:
            open(p, "w").write(s.replace('"../../include/mot.h"', '"%s"' % os.path.join(ROOT, "include", "mot.h")))
    lib = build.build(out=os.path.join(ROOT, "gpurun_out", f"libmot_ablate_{name}.so"), csrc=src, extra_flags=flags)
    if BENCH:
        import subprocess, json
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-aux", "--no-cpu-baseline"], capture_output=True, text=True,
                           env=dict(os.environ, MOT_BENCH_LIB=lib))
        try:
            d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
            print(f"{name:24s} bench {d['value']:.0f} frames/s, {d['ms_per_step']:.1f} ms/step", flush=True)
        except Exception:
            print(name, "bench failed:", r.stderr[-500:], flush=True)
        shutil.rmtree(src, ignore_errors=True); os.remove(lib)
        continue
    ctx = mot.Context(max_points=stride, max_batch=B, lib_path=lib)
    ctx.frames_dev(seq[1].data_ptr(), stride * 4, n_seq[1]); ctx.synchronize()
    print(f"{name:24s}", {i: round(ctx.time_stage(i, B, 20) * 1e3, 1) for i in ids}, "us", flush=True)
    ctx.close()
    shutil.rmtree(src, ignore_errors=True); os.remove(lib)
