"""Builds tests/emu/libmot_emu.so: the SAME csrc/*.hip sources compiled by g++ against hipemu.h.
DEVELOPMENT / CPU-TEST INFRASTRUCTURE ONLY — see hipemu.h. Never used by the product or the parity tests."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd", "csrc")
# MOT_EMU_SANITIZE=1: the same sources under UBSan (shifts, signed overflow, misaligned / out-of-bounds object accesses, …) —
#   MOT_EMU_SANITIZE=1 python -m pytest tests/test_emu_*.py tests/test_distributed_cpu.py
# float-cast-overflow is excluded: (int) of a NaN / huge float is defined on the GPU (saturating) and every such result is
# discarded by a range test before use.
SANITIZE = os.environ.get("MOT_EMU_SANITIZE", "")
# MOT_EMU_SANITIZE=address: AddressSanitizer instead — every "device" buffer is a heap allocation, so an out-of-bounds global
# load / store of a kernel (silent on the GPU until it faults) is reported with the kernel's source line. Needs the runtime
# preloaded into python:  LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 MOT_EMU_SANITIZE=address pytest ...
ASAN = SANITIZE == "address"
SAN_FLAGS = (["-fsanitize=address", "-fno-omit-frame-pointer"] if ASAN else
             ["-fsanitize=undefined", "-fno-sanitize=float-cast-overflow", "-fno-sanitize-recover=undefined"] if SANITIZE else [])
# MOT_EMU_PERTURB=1: sin / cos / exp / atan2 / pow of the kernels answer one ulp off now and then (see hipemu.h) — what the
# device math library is allowed to do; the host side of the library (mot_api.hip: the reference's libm calls) is not touched
PERTURB = bool(os.environ.get("MOT_EMU_PERTURB"))
# MOT_EMU_DEFINES="-DMOT_X=1 -DMOT_Y=2": a variant build of the kernels (the knobs tools/prebuild.py gives hipcc), in a library of its own
DEFINES = os.environ.get("MOT_EMU_DEFINES", "").split()
_TAG = ("_" + "".join(ch if ch.isalnum() else "_" for ch in "".join(DEFINES))) if DEFINES else ""
LIB = os.path.join(HERE, ("libmot_emu_asan" if ASAN else "libmot_emu_ubsan" if SANITIZE else "libmot_emu_ulp" if PERTURB else "libmot_emu") + _TAG + ".so")


def sources():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mot_build", os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd", "build.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m.SOURCES, m.HEADERS


def build(force: bool = False) -> str:
    srcs, hdrs = sources()
    deps = [os.path.join(CSRC, s) for s in srcs + hdrs] + [os.path.join(HERE, "hipemu.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    objs = []
    for s in srcs:
        o = os.path.join(HERE, ("objasan_" if ASAN else "objsan_" if SANITIZE else "objulp_" if PERTURB else "obj_") + _TAG.strip("_") + ("_" if _TAG else "") + s.replace(".hip", ".o"))
        perturb = ["-DMOT_EMU_PERTURB=1"] if PERTURB and s != "mot_api.hip" else []
        cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC"] + SAN_FLAGS + [ "-ffp-contract=off", "-fno-fast-math", "-DMOT_HIPEMU=1"] + perturb + DEFINES + [ "-x", "c++",
               "-include", os.path.join(HERE, "hipemu.h"), "-I", CSRC, "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas",
               "-Wno-unused-variable", "-c", os.path.join(CSRC, s), "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("emu build failed:\n" + r.stderr)
        objs.append(o)
    r = subprocess.run(["g++", "-shared"] + SAN_FLAGS + ["-o", LIB] + objs, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("emu link failed:\n" + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
