"""builds variants with extra -D flags and runs bench.py against each (experiment tool): variants_bench.py batch contexts name=flags ..."""
import importlib.util, os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
PKG = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd")
build = _load("mot_amd.build", os.path.join(PKG, "build.py"))
batch, ctxs = sys.argv[1], sys.argv[2]
for spec in sys.argv[3:]:
    name, _, fl = spec.partition("=")
    build.build(extra_flags=[f for f in fl.split(",") if f], force=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--batch", batch, "--contexts", ctxs], capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(name, d["value"], d["ms_per_step"], {k: round(v, 4) for k, v in d["roofline"]["kernel_ms"].items() if k in ("track_step_kernel", "cluster_index_kernel")}, flush=True)
    except Exception as e:
        print(name, "failed", r.stderr[-500:], flush=True)
build.build(force=True)
