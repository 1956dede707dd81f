"""A rosbag-free sequence player: KITTI raw `velodyne_points/data/*.bin` frames and `oxts/data/*.txt`
ego motion through the same per-frame sequence the three reference nodes run (OT/src/groundremove/main.cpp:120,
OT/src/cluster/main.cpp:74,119, OT/tracking/main.cpp:74-166) on a `Context` of this package.

`play` uses the host-buffer stage calls on purpose — it mirrors the nodes one to one; `play_fused` runs the same sequence on the
throughput path (`Context.frames_host`: pipelined uploads, one launch sequence per frame, several streams per call)."""
from __future__ import annotations

import glob
import os

import numpy as np


def load_kitti_bin(path: str) -> np.ndarray:
    """one velodyne scan: float32 x, y, z, reflectance — already the (n, 4) layout of this library"""
    a = np.fromfile(path, dtype=np.float32)
    if a.size % 4:
        raise ValueError(f"{path}: size is not a multiple of 16 bytes")
    return a.reshape(-1, 4)


def load_oxts(path: str) -> tuple[float, float]:
    """(forward velocity vf [m/s], yaw [rad]) of one KITTI oxts record (fields 8 and 5 of dataformat.txt)"""
    f = np.loadtxt(path).reshape(-1)
    return float(f[8]), float(f[5])


def kitti_frames(drive_dir: str):
    """yields (cloud, v, yaw) for every scan of a KITTI raw drive directory (…/2011_09_26_drive_0005_sync)"""
    scans = sorted(glob.glob(os.path.join(drive_dir, "velodyne_points", "data", "*.bin")))
    for s in scans:
        o = os.path.join(drive_dir, "oxts", "data", os.path.splitext(os.path.basename(s))[0] + ".txt")
        v, yaw = load_oxts(o) if os.path.exists(o) else (0.0, 0.0)
        yield load_kitti_bin(s), v, yaw


def boxes_to_global(boxes: np.ndarray, ego: np.ndarray) -> np.ndarray:
    """sensor frame -> the tracker's global frame: rotate by -egoYaw about the ego position (a plain rigid transform; the
    reference asks tf for it, OT/tracking/main.cpp:143-158)"""
    g = np.asarray(boxes, np.float64).copy()
    co, si = np.cos(-ego[2]), np.sin(-ego[2])
    dx, dy = g[..., 0] - ego[0], g[..., 1] - ego[1]
    g[..., 0] = co * dx - si * dy
    g[..., 1] = si * dx + co * dy
    return g.astype(np.float32)


def play_frame(ctx, cloud: np.ndarray, timestamp: float, v: float, yaw: float, slot: int = 0) -> dict:
    """one frame through ground removal -> clustering -> box fitting -> ego pose -> tracker, as the three nodes do"""
    g = ctx.ground_remove(cloud, want_mask=False)
    cl = ctx.cluster(g["elevated"])
    bx = ctx.box_fit(g["elevated"], cl["grid"], cl["num_cluster"])
    ego = ctx.ego_update(timestamp, v, yaw, slot)
    tr = ctx.track_step(boxes_to_global(bx["boxes"], ego), timestamp, slot)
    return dict(n_elevated=len(g["elevated"]), n_ground=len(g["ground"]), num_cluster=cl["num_cluster"], boxes=bx["boxes"], ego=ego, tracks=tr)


def play(ctx, frames, dt_us: float = 1.0e5, t0: float = 1.0e9):
    """frames: iterable of (cloud, v, yaw); timestamps advance by dt_us microseconds (the tracking node's unit)"""
    for k, (cloud, v, yaw) in enumerate(frames):
        yield play_frame(ctx, cloud, t0 + k * dt_us, v, yaw)


def play_fused(ctx, frames, dt_us: float = 1.0e5, t0: float = 1.0e9, slot_count: int = 1):
    """The same sequence on the THROUGHPUT path: one `frames_host` call per frame (pipelined upload from a page-locked staging
    block, ground -> cluster -> box -> tf -> tracker in one launch sequence, nothing but the results comes back). The change of
    frame is the tracking node's own tf chain (mot_api.hip: tf_velodyne_to_global), not the numpy transform of `play`.

    frames: iterable of (cloud, v, yaw) — or of lists of `slot_count` such triples, one per stream. Yields per frame a list of
    per-stream dicts (boxes in the sensor frame, tracks)."""
    import ctypes as C
    stride = ctx.max_points                       # points between the streams of a batch
    hp = C.c_void_p()
    ctx._ck(ctx.lib.mot_host_alloc(C.c_size_t(slot_count * stride * 16), C.byref(hp)))
    try:
        pinned = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_float)), shape=(slot_count, stride, 4))
        for k, fr in enumerate(frames):
            per = fr if isinstance(fr, list) else [fr]
            if len(per) != slot_count:
                raise ValueError(f"frame {k}: {len(per)} streams, expected {slot_count}")
            ctx.wait_uploads()                    # the previous upload has left the staging block
            n = []
            for s, (cloud, _v, _yaw) in enumerate(per):
                a = np.asarray(cloud, np.float32).reshape(-1, 4)
                if len(a) > stride:
                    raise ValueError(f"frame {k}, stream {s}: {len(a)} points, the context holds {stride}")
                pinned[s, : len(a)] = a
                n.append(len(a))
            ts = [t0 + k * dt_us] * slot_count
            ctx.frames_host(hp.value, stride * 4, n, run_tracker=True, timestamps=ts, ego_v=[p[1] for p in per], ego_yaw=[p[2] for p in per])
            yield [dict(boxes=ctx.get_boxes(s)["boxes"], tracks=ctx.get_tracks(s)) for s in range(slot_count)]
    finally:
        ctx.synchronize()
        ctx.lib.mot_host_free(hp)
