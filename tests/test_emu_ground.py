"""CPU-only development check: the ground kernels' LOGIC stepped under tests/emu/hipemu.h and compared with the
oracle. Not a parity claim (those are the -m gpu tests on the real kernels) and not a product path."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.fixture(scope="module")
def emu_ctx(mot):
    import build_emu
    lib = build_emu.build()
    c = mot.Context(lib_path=lib, max_points=16384, max_batch=2)
    yield c
    c.close()


@pytest.mark.parametrize("n,stream", [(0, 0), (1, 0), (2049, 1), (9000, 2)])
def test_emu_ground(emu_ctx, oracle, synth, n, stream):
    p = oracle.params(0)
    cloud = synth.make_cloud(max(n, 1), stream, 0)[:n]
    if n > 100:
        cloud = np.concatenate([cloud, synth.edge_case_points()])
    r = emu_ctx.ground_remove(cloud)
    g = oracle.ground_remove(p, cloud)
    assert np.array_equal(r["mask"], g["mask"])
    assert np.array_equal(r["elevated"], g["elevated"]) and np.array_equal(r["ground"], g["ground"])


def test_emu_decode_pointcloud2(emu_ctx, synth):
    """the PointCloud2 -> float4 kernel under the emulator ("device" pointers are host arrays there)"""
    n, step = 5000, 22
    cloud = synth.make_cloud(n, 2, 0)
    raw = np.random.default_rng(1).integers(0, 256, size=(n, step), dtype=np.uint8)
    for k, off in enumerate((0, 4, 8, 12)):
        raw[:, off:off + 4] = cloud[:, k].copy().view(np.uint8).reshape(n, 4)
    out = np.zeros((n, 4), np.float32)
    emu_ctx.decode_pointcloud2_dev(raw.ctypes.data, n, step, 0, 4, 8, 12, out.ctypes.data)
    emu_ctx.synchronize()
    assert np.array_equal(out, cloud)
