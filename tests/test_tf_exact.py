"""the sensor -> global change of frame INSIDE the fused device path must be the tracking node's
(OT/tracking/main.cpp:76-83,143-158: tf broadcast + pcl_ros::transformPointCloud per box), not a formula of our own.
Expected values come from the reference node's own call sequence executed on the tf / pcl_ros code the node-level oracle
runs on (oracle/ref_tf_capi.cpp in oracle/_ref/libmot_ref.so) and from the golden fixture generated from it
(tests/golden/tf_boxes.npz, tests/golden/make_golden.py). Bit-exact."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import golden_util as G

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


def _apply(m, boxes):
    """pcl::transformPointCloud's arithmetic: fp32, left to right"""
    m = np.asarray(m, np.float32).reshape(3, 4); b = np.asarray(boxes, np.float32)
    x, y, z = b[..., 0], b[..., 1], b[..., 2]
    return np.stack([((m[r, 0] * x + m[r, 1] * y).astype(np.float32) + m[r, 2] * z).astype(np.float32) + m[r, 3] for r in range(3)], -1).astype(np.float32)


def _matrix(lib, x, y, yaw):
    m = np.zeros(12, np.float32)
    assert lib.mot_debug_tf_matrix(C.c_double(x), C.c_double(y), C.c_double(yaw), m.ctypes.data_as(C.c_void_p)) == 0
    return m


def _ref_boxes(oracle, boxes, pose):
    out = np.zeros_like(boxes)
    rc = oracle.ref().ref_boxes_to_global(boxes.ctypes.data_as(C.c_void_p), len(boxes), C.c_double(pose[0]), C.c_double(pose[1]), C.c_double(pose[2]),
                                          out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out


def test_matrix_against_golden_fixture(mot):
    import build_emu
    lib = mot.load_library(build_emu.build())
    fx = G.load("tf_boxes.npz")
    for pose, boxes, want in zip(fx["pose"], fx["boxes"], fx["global"]):
        got = _apply(_matrix(lib, *pose), boxes)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), pose


def test_matrix_against_the_reference_call_sequence(mot, oracle):
    if oracle.ref() is None:
        pytest.skip("oracle/_ref not built")
    import build_emu
    lib = mot.load_library(build_emu.build())
    rng = np.random.default_rng(3)
    for k in range(3000):
        yaw = rng.uniform(-7, 7) if k % 5 else rng.choice([0.0, np.pi, -np.pi, np.pi / 2, -np.pi / 2, 3.0, -3.1415926])
        pose = (rng.uniform(-300, 300), rng.uniform(-300, 300), yaw)
        boxes = rng.uniform(-60, 60, size=(4, 8, 3)).astype(np.float32)
        want = _ref_boxes(oracle, boxes, pose)
        got = _apply(_matrix(lib, *pose), boxes)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (k, pose)


def _fused_boxes(mot, synth, oracle, lib_path, hip_lib=None):
    """frames through the fused path with a moving ego; returns [(pose, sensor boxes, global boxes of the device)]"""
    N, stride = 12000, 12288
    kw = dict(lib_path=lib_path) if lib_path else {}
    out = []
    with mot.Context(max_points=stride, max_batch=1, max_tracks_total=256, **kw) as c, mot.Context(max_points=64, **kw) as shadow:
        L = c.lib
        import hiprt
        for f in range(5):
            cloud = np.zeros((stride, 4), np.float32); cloud[:N] = synth.make_cloud(N, 3, f)
            ts = 1.0e9 + f * 1e5; v, yaw = 3.0 + f, 0.03 * f - 0.5
            if lib_path:
                c.frames_dev(cloud.ctypes.data, stride * 4, [N], run_tracker=True, timestamps=[ts], ego_v=[v], ego_yaw=[yaw])
            else:
                dev = hiprt.DeviceBuffer(cloud)
                c.frames_dev(dev.ptr, stride * 4, [N], run_tracker=True, timestamps=[ts], ego_v=[v], ego_yaw=[yaw])
            bx = c.get_boxes(0)["boxes"]
            pose = shadow.ego_update(ts, v, yaw)[:3]       # the same dead reckoning (host libm), on a second context
            shadow.track_step(np.zeros((0, 8, 3), np.float32), ts)   # advances the shadow's timestamp_ / egoPreYaw_
            g = np.zeros((1024, 8, 3), np.float32)
            assert L.mot_debug_copy(c._h, 11, 0, g.ctypes.data_as(C.c_void_p), C.c_size_t(g.nbytes)) == 0
            out.append((pose, bx, g[: len(bx)].copy()))
    return out


def test_fused_path_boxes_equal_the_nodes_tf_path_emulated(mot, synth, oracle):
    if oracle.ref() is None:
        pytest.skip("oracle/_ref not built")
    import build_emu
    n = 0
    for pose, bx, got in _fused_boxes(mot, synth, oracle, build_emu.build()):
        want = _ref_boxes(oracle, bx, pose)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), pose
        n += len(bx)
    assert n > 5


@pytest.mark.gpu
def test_fused_path_boxes_equal_the_nodes_tf_path(mot, hip_lib, synth, oracle):
    n = 0
    for pose, bx, got in _fused_boxes(mot, synth, oracle, None):
        if oracle.ref() is not None:
            want = _ref_boxes(oracle, bx, pose)
        else:
            want = _apply(_matrix(hip_lib, *pose), bx)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), pose
        n += len(bx)
    assert n > 5
    fx = G.load("tf_boxes.npz")   # and the host chain of the real library against the fixture generated from the reference's sequence
    for pose, boxes, want in zip(fx["pose"], fx["boxes"], fx["global"]):
        assert np.array_equal(_apply(_matrix(hip_lib, *pose), boxes).view(np.uint32), want.view(np.uint32))
