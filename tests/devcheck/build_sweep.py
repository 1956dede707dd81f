"""tests/devcheck/sweep.hip -> libmot_sweep.so (hipcc, gfx950) or, for the CPU suite, libmot_sweep_emu.so (g++ against tests/emu/hipemu.h).
TEST INFRASTRUCTURE: the on-device sweeps of the guarded fast cell paths; not part of libmot_hip.so."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd", "csrc")
SRC = os.path.join(HERE, "sweep.hip")
LIB = os.path.join(HERE, "libmot_sweep.so")
LIB_EMU = os.path.join(HERE, "libmot_sweep_emu.so")


def _stale(lib):
    deps = [SRC, os.path.join(CSRC, "mot_internal.h"), os.path.join(CSRC, "mot_math.h"), os.path.abspath(__file__)]
    return not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps)


def build(force=False):
    if not force and not _stale(LIB):
        return LIB
    hipcc = next((c for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc") if c and os.path.exists(c)), None)
    if hipcc is None:
        raise RuntimeError("hipcc not found")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
           "-fno-gpu-flush-denormals-to-zero", "-I", CSRC, SRC, "-o", LIB + ".tmp"]   # the product's flags: the same device functions, compiled the same way
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


def build_emu(force=False):
    if not force and not _stale(LIB_EMU):
        return LIB_EMU
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-DMOT_HIPEMU=1", "-x", "c++", "-include",
           os.path.join(ROOT, "tests", "emu", "hipemu.h"), "-I", CSRC, "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-unused-variable", SRC, "-o", LIB_EMU]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr)
    return LIB_EMU


if __name__ == "__main__":
    print(build(force=True))
