"""Builds variant libraries HERE (hipcc cross-compiles without a GPU) so that a `gpurun` call spends its minutes measuring, not
compiling:  python tools/prebuild.py [-j4] name=-DFLAG[,-DFLAG...] ... | ablate:<experiment of tools/ablate.py>
Outputs variants/libmot_<name>.so (git-ignored, travels with the snapshot); tools/bench_variants.sh runs bench.py on each.
Experiment tooling, not product code."""
import concurrent.futures as cf, importlib.util, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd")
def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
build = _load("mot_amd.build", os.path.join(PKG, "build.py"))
OUT = os.path.join(ROOT, "variants")

def one(spec):
    name, _, fl = spec.partition("=")
    flags = [f for f in fl.split(",") if f]
    out = os.path.join(OUT, f"libmot_{name}.so")
    build.build(extra_flags=flags, out=out)
    return name, flags

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    jobs = 4
    specs = []
    for a in sys.argv[1:]:
        if a.startswith("-j"): jobs = int(a[2:])
        else: specs.append(a)
    with cf.ThreadPoolExecutor(jobs) as ex:
        for name, flags in ex.map(one, specs):
            print("built", name, " ".join(flags), flush=True)
