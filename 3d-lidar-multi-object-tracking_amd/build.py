"""Builds the HIP extension in-tree:  csrc/*.hip  ->  libmot_hip.so  (gfx950 only).

hipcc cross-compiles without a GPU, so this also is the "does it build" check of __graft_entry__.build().
Flags that matter for parity with the reference's x86-64 build (no FMA, IEEE divide/sqrt):
  -ffp-contract=off  -fhip-fp32-correctly-rounded-divide-sqrt  (denormals are kept: no flush-to-zero flag)
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmot_hip.so")
SOURCES = ["ground.hip", "cluster.hip", "box.hip", "side.hip", "track.hip", "mot_api.hip"]
HEADERS = ["mot_internal.h", "mot_math.h", "mot_wave.h", "mot_debug.h", "mot_debug_api.h", "mot_track_prep.h", os.path.join("..", "..", "include", "mot.h")]

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
    "-fno-gpu-flush-denormals-to-zero",
    "-Wall", "-Wno-unused-function",
]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str | None = None, csrc: str | None = None) -> str:
    """extra_flags / out / csrc: experiment builds (geometry knobs, timing stamps, patched scratch copies of the sources for
    ablations — tools/ablate.py), written next to gpurun_out/, never the product"""
    if out is None and not force and not needs_build():
        return LIB
    out = out or LIB
    cmd = [hipcc()] + HIPCC_FLAGS + list(extra_flags) + [os.path.join(csrc or CSRC, s) for s in SOURCES] + ["-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    if verbose and r.stderr:
        print(r.stderr, file=sys.stderr)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
