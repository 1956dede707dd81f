"""ROS1 wire format in ~100 lines, and the message logs of the file-backed mini-ROS (oracle/ref_shim/ros/ros.h).

TEST INFRASTRUCTURE. The node executables under test (the reference's own main.cpp files built into oracle/_ref/bin, and this
repository's ros/src/*.cpp) are run as `<node> --in IN.log --out OUT.log`; this module writes IN.log, reads OUT.log and
decodes the payloads, which are genuine ROS1 serialisations (little endian, fields in declaration order, uint32 counts in
front of strings and variable-length arrays, time = 2 x uint32, duration = 2 x int32). The message definitions are the
public ROS ones (std_msgs, sensor_msgs, geometry_msgs, nav_msgs, visualization_msgs) and the package's four custom types
(field lists as in OT/msg/*.msg)."""
from __future__ import annotations

import struct

import numpy as np

SPECS = {
    "std_msgs/Header": "uint32 seq|time stamp|string frame_id",
    "std_msgs/ColorRGBA": "float32 r|float32 g|float32 b|float32 a",
    "geometry_msgs/Point": "float64 x|float64 y|float64 z",
    "geometry_msgs/Vector3": "float64 x|float64 y|float64 z",
    "geometry_msgs/Quaternion": "float64 x|float64 y|float64 z|float64 w",
    "geometry_msgs/Pose": "geometry_msgs/Point position|geometry_msgs/Quaternion orientation",
    "geometry_msgs/Twist": "geometry_msgs/Vector3 linear|geometry_msgs/Vector3 angular",
    "geometry_msgs/PoseWithCovariance": "geometry_msgs/Pose pose|float64[36] covariance",
    "geometry_msgs/TwistWithCovariance": "geometry_msgs/Twist twist|float64[36] covariance",
    "sensor_msgs/PointField": "string name|uint32 offset|uint8 datatype|uint32 count",
    "sensor_msgs/PointCloud2": "std_msgs/Header header|uint32 height|uint32 width|sensor_msgs/PointField[] fields|bool is_bigendian|"
                               "uint32 point_step|uint32 row_step|uint8[] data|bool is_dense",
    "nav_msgs/MapMetaData": "time map_load_time|float32 resolution|uint32 width|uint32 height|geometry_msgs/Pose origin",
    "nav_msgs/OccupancyGrid": "std_msgs/Header header|nav_msgs/MapMetaData info|int8[] data",
    "nav_msgs/Odometry": "std_msgs/Header header|string child_frame_id|geometry_msgs/PoseWithCovariance pose|"
                         "geometry_msgs/TwistWithCovariance twist",
    "visualization_msgs/Marker": "std_msgs/Header header|string ns|int32 id|int32 type|int32 action|geometry_msgs/Pose pose|"
                                 "geometry_msgs/Vector3 scale|std_msgs/ColorRGBA color|duration lifetime|bool frame_locked|"
                                 "geometry_msgs/Point[] points|std_msgs/ColorRGBA[] colors|string text|string mesh_resource|"
                                 "bool mesh_use_embedded_materials",
    "visualization_msgs/MarkerArray": "visualization_msgs/Marker[] markers",
    "object_tracking/Obstacle": "float64 x|float64 y|float64 z|float64 yaw|float64 pitch|float64 roll|int32 cluster|float64 speed",
    "object_tracking/ObstacleList": "std_msgs/Header header|float64 cellLength|float64 cellWidth|object_tracking/Obstacle[] obstacles",
    "object_tracking/trackbox": "std_msgs/Header header|uint8 box_num|float32[] x1|float32[] x2|float32[] x3|float32[] x4|"
                                "float32[] y1|float32[] y2|float32[] y3|float32[] y4",
}
PRIM = {"bool": "u1", "uint8": "u1", "int8": "i1", "uint16": "<u2", "int16": "<i2", "uint32": "<u4", "int32": "<i4", "uint64": "<u8",
        "int64": "<i8", "float32": "<f4", "float64": "<f8"}


def _fields(typ):
    for f in SPECS[typ].split("|"):
        t, name = f.split(" ")
        n = None
        if t.endswith("]"):
            t, dim = t[:-1].split("[")
            n = int(dim) if dim else -1
        yield t, name, n


def _dec(typ, buf, o):
    if typ in PRIM:
        dt = np.dtype(PRIM[typ]); return np.frombuffer(buf, dt, 1, o)[0].item(), o + dt.itemsize
    if typ == "string":
        n = struct.unpack_from("<I", buf, o)[0]; return bytes(buf[o + 4:o + 4 + n]).decode(), o + 4 + n
    if typ in ("time", "duration"):
        s, ns = struct.unpack_from("<II" if typ == "time" else "<ii", buf, o); return (s, ns), o + 8
    out = {}
    for t, name, n in _fields(typ):
        if n is None:
            out[name], o = _dec(t, buf, o); continue
        if n < 0:
            n = struct.unpack_from("<I", buf, o)[0]; o += 4
        if t in PRIM:
            dt = np.dtype(PRIM[t]); out[name] = np.frombuffer(buf, dt, n, o).copy(); o += dt.itemsize * n
        else:
            items = []
            for _ in range(n):
                v, o = _dec(t, buf, o); items.append(v)
            out[name] = items
    return out, o


def decode(typ: str, data: bytes) -> dict:
    v, o = _dec(typ, memoryview(data), 0)
    assert o == len(data), (typ, o, len(data))
    return v


def _enc(typ, v, out):
    if typ in PRIM:
        out.append(np.array(v, np.dtype(PRIM[typ])).tobytes()); return
    if typ == "string":
        b = v.encode(); out.append(struct.pack("<I", len(b)) + b); return
    if typ in ("time", "duration"):
        out.append(struct.pack("<II" if typ == "time" else "<ii", *v)); return
    for t, name, n in _fields(typ):
        x = v.get(name)
        if n is None:
            _enc(t, x if x is not None else _default(t), out); continue
        x = [] if x is None else x
        if n < 0:
            out.append(struct.pack("<I", len(x)))
        elif len(x) == 0:
            x = [_default(t)] * n
        if t in PRIM:
            out.append(np.ascontiguousarray(x, np.dtype(PRIM[t])).tobytes())
        else:
            for item in x:
                _enc(t, item, out)


def _default(t):
    return 0 if t in PRIM else "" if t == "string" else (0, 0) if t in ("time", "duration") else {}


def encode(typ: str, v: dict) -> bytes:
    out = []; _enc(typ, v, out); return b"".join(out)


def stamp(t: float):
    """ros::Time::fromSec"""
    s = int(np.floor(t)); ns = int(round((t - s) * 1e9)); return (s + ns // 1000000000, ns % 1000000000)


# ------------------------------------------------------------------ logs
def write_log(path, records):
    """records: iterable of (topic, type, payload bytes | dict); ("__now__", t) sets ros::Time::now()"""
    with open(path, "wb") as f:
        for r in records:
            if r[0] == "__now__":
                topic, typ, data = "__now__", "float64", struct.pack("<d", r[1])
            else:
                topic, typ, data = r
                if isinstance(data, dict):
                    data = encode(typ, data)
            for s in (topic.encode(), typ.encode(), data):
                f.write(struct.pack("<I", len(s))); f.write(s)


def read_log(path):
    """-> list of (topic, type, payload bytes)"""
    buf = open(path, "rb").read(); o = 0; out = []
    while o < len(buf):
        rec = []
        for _ in range(3):
            n = struct.unpack_from("<I", buf, o)[0]; rec.append(buf[o + 4:o + 4 + n]); o += 4 + n
        out.append((rec[0].decode(), rec[1].decode(), rec[2]))
    return out


def pointcloud2(xyzi: np.ndarray, t: float, frame_id="velodyne", seq=0, point_step=16, names=("x", "y", "z", "intensity")) -> dict:
    """the PointCloud2 kitti2bag / the velodyne driver publish: float32 fields at 4-byte offsets, one row"""
    a = np.ascontiguousarray(xyzi, np.float32); n = len(a)
    raw = np.zeros((n, point_step), np.uint8)
    raw[:, :4 * a.shape[1]] = a.view(np.uint8).reshape(n, 4 * a.shape[1])
    fields = [dict(name=nm, offset=4 * k, datatype=7, count=1) for k, nm in enumerate(names[: a.shape[1]])]
    return dict(header=dict(seq=seq, stamp=stamp(t), frame_id=frame_id), height=1, width=n, fields=fields, is_bigendian=0,
                point_step=point_step, row_step=point_step * n, data=raw.reshape(-1), is_dense=1)


def cloud_xyz(msg: dict) -> np.ndarray:
    """x,y,z,(4th float) of a decoded PointCloud2 with 16-byte points"""
    assert msg["point_step"] == 16
    return msg["data"].view(np.float32).reshape(-1, 4)
