"""Builds the experiment probes under tools/probes into variants/ (git-ignored; travels to the GPU box with the snapshot)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd", "csrc")
OUT = os.path.join(ROOT, "variants")


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, "tools", "probes", "frame_team_probe.hip")
    lib = os.path.join(OUT, "libframe_team_probe.so")
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= os.path.getmtime(src):
        return lib
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-fno-gpu-flush-denormals-to-zero", "-I", CSRC, src, "-o", lib]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
