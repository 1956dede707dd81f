"""Builds the node executables that tests/test_nodes.py compares (TEST INFRASTRUCTURE, CPU only):

  reference  oracle/_ref/bin/{ground,cluster,tracking}      the reference's main.cpp + its own algorithm sources (oracle/Makefile)
  recipe     oracle/_ref/bin/recipe_{ground,cluster,tracking}   the reference's main.cpp with the ONE edit INTEGRATION.md prescribes
             (the include of the algorithm header replaced by "mot_adapters.hpp"; the edited text goes to the compiler through
             a pipe — no copy of a reference source is ever written into this repository), linked against the C-ABI library
  own        tests/emu/bin/{ground,cluster,tracking}        this repository's ros/src/*_node.cpp, linked against the C-ABI library

All of them are compiled against the file-backed mini-ROS of oracle/ref_shim and run as `<node> --in IN.log --out OUT.log`.
The C-ABI library is the CPU emulator build (tests/emu) here; on a ROS machine the same sources link libmot_hip.so."""
from __future__ import annotations

import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/object_tracking"
SHIM = os.path.join(ROOT, "oracle", "ref_shim")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "bin")
OWN_BIN = os.path.join(HERE, "emu", "bin")
HIP_BIN = os.path.join(ROOT, "ros", "bin")     # the same node sources linked against libmot_hip.so: prebuilt here, run on the GPU box
HIP_LIB = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd", "libmot_hip.so")
NODES = {"ground": "src/groundremove/main.cpp", "cluster": "src/cluster/main.cpp", "tracking": "tracking/main.cpp"}
ALGO_HEADERS = ("ground_removal.h", "component_clustering.h", "box_fitting.h", "imm_ukf_jpda.h")
FLAGS = ["-std=c++14", "-O2", "-ffp-contract=off", "-fno-fast-math", "-w"]


def have_reference() -> bool:
    return os.path.isfile(os.path.join(REF, "tracking", "main.cpp"))


def _newer(target, deps, lib=None):
    """up to date w.r.t. deps AND linked against the same library as last time (emulator vs its UBSan build vs libmot_hip.so)"""
    stamp = target + ".lib"
    if lib is not None and (not os.path.exists(stamp) or open(stamp).read() != lib):
        return False
    return os.path.exists(target) and all(os.path.getmtime(d) <= os.path.getmtime(target) for d in deps if os.path.exists(d))


def _stamp(target, lib):
    open(target + ".lib", "w").write(lib)


def _shim_files():
    return [os.path.join(d, f) for d, _, fs in os.walk(SHIM) for f in fs]


def _link_args(lib):
    d, f = os.path.split(lib)
    return ["-L", d, "-l:" + f, "-Wl,-rpath," + d]


def _cxx(args, what, stdin_source=None):
    r = subprocess.run(["g++"] + FLAGS + args, capture_output=True, text=True, input=stdin_source)
    if r.returncode:
        raise RuntimeError(f"building {what} failed:\n{r.stderr[-4000:]}")


def reference_nodes() -> dict:
    """the three nodes of OT/ plus `pipeline0`, the single-process node of OT0/ (OT0/src/main.cpp)"""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, capture_output=True)
    return {n: os.path.join(REF_BIN, n) for n in list(NODES) + ["pipeline0"]}


def recipe_nodes(lib: str, out_dir: str = REF_BIN, link_extra=()) -> dict:
    """the reference's node sources with only their algorithm includes swapped for the adapter header"""
    os.makedirs(out_dir, exist_ok=True)
    out = {}
    pat = re.compile(r'^#include "(%s)"\s*$' % "|".join(re.escape(h) for h in ALGO_HEADERS), re.M)
    for n, rel in NODES.items():
        path = os.path.join(REF, rel)
        exe = os.path.join(out_dir, "recipe_" + n)
        deps = [path, lib, os.path.join(ROOT, "include", "mot_adapters.hpp"), os.path.join(ROOT, "include", "mot.h"), os.path.abspath(__file__)] + _shim_files()
        if not _newer(exe, deps, lib):
            src = open(path, encoding="utf-8", errors="replace").read()
            assert pat.search(src), rel
            first = [True]
            def swap(m):
                if first[0]:
                    first[0] = False; return '#include "mot_adapters.hpp"'
                return ""
            _cxx(["-I", SHIM, "-I", os.path.join(REF, "tracking"), "-I", os.path.join(ROOT, "include"), "-x", "c++", "-", "-o", exe] + _link_args(lib) + list(link_extra),
                 exe, stdin_source=pat.sub(swap, src))
            _stamp(exe, lib)
        out[n] = exe
    return out


def own_nodes(lib: str, out_dir: str = OWN_BIN, link_extra=()) -> dict:
    """ros/src/<node>_node.cpp of this repository"""
    os.makedirs(out_dir, exist_ok=True)
    out = {}
    common = [os.path.join(ROOT, "ros", "src", f) for f in os.listdir(os.path.join(ROOT, "ros", "src")) if f.endswith((".hpp", ".h"))]
    for n in list(NODES) + ["pipeline"]:
        src = os.path.join(ROOT, "ros", "src", f"{n}_node.cpp")
        exe = os.path.join(out_dir, n)
        deps = [src, lib, os.path.join(ROOT, "include", "mot.h")] + common + _shim_files()
        if not _newer(exe, deps, lib):
            _cxx(["-I", SHIM, "-I", os.path.join(REF, "tracking"), "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "ros", "src"),
                  src, "-o", exe] + _link_args(lib) + list(link_extra), exe)
            _stamp(exe, lib)
        out[n] = exe
    return out


def adapter_driver(lib: str, out_dir: str = OWN_BIN, link_extra=()) -> str:
    """tests/drivers/adapter_tracker_driver.cpp: the adapter header's tracker functions on a small track budget"""
    os.makedirs(out_dir, exist_ok=True)
    src = os.path.join(HERE, "drivers", "adapter_tracker_driver.cpp")
    exe = os.path.join(out_dir, "adapter_tracker_driver")
    deps = [src, lib, os.path.join(ROOT, "include", "mot_adapters.hpp"), os.path.join(ROOT, "include", "mot.h")] + _shim_files()
    if not _newer(exe, deps, lib):
        _cxx(["-I", SHIM, "-I", os.path.join(REF, "tracking"), "-I", os.path.join(ROOT, "include"), src, "-o", exe] + _link_args(lib) + list(link_extra), exe)
        _stamp(exe, lib)
    return exe


def _hip_link_extra():
    rel = os.path.relpath(os.path.dirname(HIP_LIB), HIP_BIN)
    return ["-Wl,-rpath,$ORIGIN/" + rel, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]


def hip_recipe_nodes() -> dict:
    """the reference's UNMODIFIED node sources + the adapter header (recipe_nodes) linked against the real library: ros/bin/recipe_*,
    built where /root/reference exists, run on the GPU box (tests/test_nodes_gpu.py). Plus the adapter's tracker driver."""
    os.makedirs(HIP_BIN, exist_ok=True)
    out = recipe_nodes(HIP_LIB, HIP_BIN, _hip_link_extra())
    out["adapter_tracker_driver"] = adapter_driver(HIP_LIB, HIP_BIN, _hip_link_extra())
    return out


def hip_nodes() -> dict:
    """the node shells linked against the real library (needs the reference's vendored Eigen for the tf / pcl_ros shim, so it
    is built in the container that has /root/reference — __graft_entry__.build() — and travels to the GPU box prebuilt).
    The run path is relative to the executable, the HIP runtime is found through the library's own run path."""
    return own_nodes(HIP_LIB, HIP_BIN, _hip_link_extra())


def prebuilt_recipe(dirname: str = HIP_BIN):
    """{node: path} of the recipe executables (+ the adapter driver) when they all exist in dirname, else None"""
    out = {n: os.path.join(dirname, "recipe_" + n) for n in NODES}
    out["adapter_tracker_driver"] = os.path.join(dirname, "adapter_tracker_driver")
    return out if all(os.path.isfile(p) and os.access(p, os.X_OK) for p in out.values()) else None


def prebuilt(dirname: str):
    """{node: path} when all three executables exist in dirname, else None"""
    out = {n: os.path.join(dirname, n) for n in list(NODES) + ["pipeline0" if dirname == REF_BIN else "pipeline"]}
    return out if all(os.path.isfile(p) and os.access(p, os.X_OK) for p in out.values()) else None


def run_node(exe: str, in_log: str, out_log: str, params: dict | None = None, timeout: int = 600, cwd: str | None = None):
    args = [exe, "--in", in_log, "--out", out_log] + [f"{k}:={v}" for k, v in (params or {}).items()]
    r = subprocess.run(args, capture_output=True, text=True, timeout=timeout, errors="replace", cwd=cwd)
    if r.returncode:
        raise RuntimeError(f"{exe} exited with {r.returncode}:\n{r.stderr[-2000:]}")
    return r
