"""The two independent ROS1 wire codecs of the test infrastructure agree: random messages of every type the nodes exchange,
encoded by tests/roslog.py (generic, driven by the message definitions), must be read and written back byte for byte by the
hand-written C++ codecs of oracle/ref_shim (through oracle/ref_shim/echo_node.cpp); plus known-answer layouts."""
import os
import struct
import subprocess

import numpy as np

import roslog as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TYPES = ("sensor_msgs/PointCloud2", "nav_msgs/OccupancyGrid", "nav_msgs/Odometry", "object_tracking/ObstacleList",
         "object_tracking/trackbox", "visualization_msgs/Marker", "visualization_msgs/MarkerArray")


def random_value(rng, t, n=None):
    if n is not None:
        k = n if n >= 0 else int(rng.integers(0, 5))
        if t in R.PRIM:
            return random_prims(rng, t, k)
        return [random_value(rng, t) for _ in range(k)]
    if t in R.PRIM:
        return random_prims(rng, t, 1)[0].item()
    if t == "string":
        return "".join(chr(int(c)) for c in rng.integers(32, 127, size=int(rng.integers(0, 12))))
    if t == "time":
        return (int(rng.integers(0, 2**32)), int(rng.integers(0, 10**9)))
    if t == "duration":
        return (int(rng.integers(-2**31, 2**31)), int(rng.integers(0, 10**9)))
    return {name: random_value(rng, ft, fn) for ft, name, fn in R._fields(t)}


def random_prims(rng, t, k):
    dt = np.dtype(R.PRIM[t])
    if t == "bool":
        return rng.integers(0, 2, size=k).astype(dt)
    if dt.kind == "f":
        return rng.standard_normal(k).astype(dt)
    info = np.iinfo(dt)
    return rng.integers(info.min, int(info.max) + 1, size=k, dtype=np.int64 if dt.kind == "i" else np.uint64).astype(dt)


def test_python_codec_known_answers():
    h = dict(seq=7, stamp=(3, 5), frame_id="ab")
    assert R.encode("std_msgs/Header", h) == struct.pack("<III", 7, 3, 5) + struct.pack("<I", 2) + b"ab"
    tb = dict(header=h, box_num=1, x1=[1.0, 2.0, 3.0])
    b = R.encode("object_tracking/trackbox", tb)
    assert b[18:19] == b"\x01" and b[19:23] == struct.pack("<I", 3) and len(b) == 18 + 1 + 4 + 12 + 7 * 4
    assert R.decode("object_tracking/trackbox", b)["x1"].tolist() == [1.0, 2.0, 3.0]
    assert R.stamp(1000.25) == (1000, 250000000)


def test_cxx_codecs_match_python_codec(tmp_path):
    exe = str(tmp_path / "echo")
    eigen = "/root/reference/object_tracking/tracking"
    inc = ["-I", os.path.join(ROOT, "oracle", "ref_shim")] + (["-I", eigen] if os.path.isdir(eigen) else [])
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-w"] + inc + [os.path.join(ROOT, "oracle", "ref_shim", "echo_node.cpp"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    rng = np.random.default_rng(11)
    recs = []
    for t in TYPES:
        for _ in range(8):
            recs.append(("echo/" + t, t, R.encode(t, random_value(rng, t))))
    R.write_log(str(tmp_path / "in.log"), recs)
    subprocess.run([exe, "--in", str(tmp_path / "in.log"), "--out", str(tmp_path / "out.log")], check=True, capture_output=True)
    out = R.read_log(str(tmp_path / "out.log"))
    assert len(out) == len(recs)
    for (t, ty, b), (t2, ty2, b2) in zip(recs, out):
        assert (t, ty) == (t2, ty2) and b == b2, t
        assert R.encode(ty, R.decode(ty, b)) == b


def test_tf_shim_identities(tmp_path):
    """the restated tf / pcl_ros slice (oracle/ref_shim/tf, pcl_ros): rotation matrices, quaternion round trips, both lookup
    directions of the node's one edge, ros::Time / Duration conversions"""
    import pytest
    eigen = "/root/reference/object_tracking/tracking"
    if not os.path.isdir(eigen):
        pytest.skip("needs the reference's vendored Eigen")
    exe = str(tmp_path / "selftest_tf")
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-ffp-contract=off", "-w", "-I", os.path.join(ROOT, "oracle", "ref_shim"), "-I", eigen,
                        os.path.join(ROOT, "oracle", "ref_shim", "selftest_tf.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout[-2000:]
