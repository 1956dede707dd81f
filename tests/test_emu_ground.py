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


def test_emu_ground_remove_pointcloud2_and_resident_box_fit(mot, oracle, synth):
    """the two entry points the node shells use: raw PointCloud2 records in (unpacked on the device, pre-filter fused), and
    the box stage on what the cluster stage left resident"""
    import build_emu
    lib = build_emu.build()
    n, step = 9000, 32
    cloud = np.concatenate([synth.make_cloud(n, 1, 0), synth.edge_case_points()]); n = len(cloud)
    raw = np.random.default_rng(2).integers(0, 256, size=(n, step), dtype=np.uint8)
    for k, off in enumerate((4, 8, 16)):
        raw[:, off:off + 4] = cloud[:, k].copy().view(np.uint8).reshape(n, 4)
    for crop in (0, 1):
        p = oracle.params(0, crop_enable=crop)
        with mot.Context(mot.params(0, lib=mot.load_library(lib), crop_enable=crop), lib_path=lib, max_points=16384) as c:
            r = c.ground_remove_pointcloud2(raw, n, step, 4, 8, 16)
            g = oracle.ground_remove(p, oracle.crop(p, cloud) if crop else cloud)
            assert np.array_equal(r["elevated"][:, :3], g["elevated"][:, :3]) and np.array_equal(r["ground"][:, :3], g["ground"][:, :3])
            assert np.all(r["elevated"][:, 3] == 1.0) and np.all(r["ground"][:, 3] == 1.0)   # pcl::PointXYZ's padding
            assert np.array_equal(c.ground_remove_pointcloud2(raw, 0, step, 4, 8, 16)["elevated"], np.zeros((0, 4), np.float32))
            cl = c.cluster(r["elevated"])
            bx = c.box_fit_resident()
            ob = oracle.box_fit(p, g["elevated"], oracle.cluster(p, g["elevated"])["grid"], cl["num_cluster"])
            assert np.array_equal(bx["boxes"], ob["boxes"]) and len(bx["boxes"]) > 0
            assert np.array_equal(bx["boxes"], c.box_fit(r["elevated"], cl["grid"], cl["num_cluster"])["boxes"])
            c.frame_pointcloud2(raw, n, step, 4, 8, 16)     # the three stages in one call, cloud resident throughout
            assert np.array_equal(c.get_boxes(0)["boxes"], ob["boxes"])
            assert np.array_equal(c.get_ground(0)["elevated"][:, :3], g["elevated"][:, :3])


@pytest.mark.parametrize("cap", [1, 63, 64, 65, 4095, 4096, 4097, 8191])
def test_emu_frames_that_fill_the_capacity_exactly(mot, oracle, synth, cap):
    """n == max_points for capacities around the 64-point and 4096-point granularities of the kernels (under
    MOT_EMU_SANITIZE=address an out-of-bounds access of the padded tails would be reported)"""
    import build_emu
    lib = build_emu.build()
    p = oracle.params(0)
    cloud = synth.make_cloud(max(cap, 64), 6, 0)[:cap]
    with mot.Context(lib_path=lib, max_points=cap, max_batch=2, max_tracks_total=64) as c:
        r = c.ground_remove(cloud); g = oracle.ground_remove(p, cloud)
        assert np.array_equal(r["mask"], g["mask"]) and np.array_equal(r["elevated"], g["elevated"]) and np.array_equal(r["ground"], g["ground"])
        e = np.repeat(cloud[: max(cap // 4, 1)], 4, axis=0)[:cap]; e[:, 2] = 0.0   # an elevated cloud that also fills the capacity
        cl = c.cluster(e); ocl = oracle.cluster(p, e)
        assert cl["num_cluster"] == ocl["num_cluster"] and np.array_equal(cl["grid"], ocl["grid"]) and np.array_equal(cl["point_label"], ocl["point_label"])
        assert np.array_equal(c.box_fit_resident()["boxes"], oracle.box_fit(p, e, ocl["grid"], ocl["num_cluster"])["boxes"])
        host = np.zeros((2, c.max_points if hasattr(c, "max_points") else cap, 4), np.float32)
        stride = host.shape[1]
        host[0, :cap] = cloud; host[1, :cap] = cloud[::-1]
        c.frames_dev(host.ctypes.data, stride * 4, [cap, cap])
        assert np.array_equal(c.get_ground(0)["elevated"], g["elevated"])
        assert np.array_equal(c.get_ground(1)["elevated"], oracle.ground_remove(p, cloud[::-1])["elevated"])
