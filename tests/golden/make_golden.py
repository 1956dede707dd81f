"""Generates tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref = the reference's own sources compiled
against the shim, see oracle/Makefile). Run in the build container, where /root/reference exists:

    python tests/golden/make_golden.py

The reference ships no golden vectors of its own, so these fixtures are the pinned record of what
the reference code computes on fixed seeded inputs; tests compare the C restatement and the HIP path to them.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util

import oracle_lib as O

spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "tools", "synth", "synth.py"))
S = importlib.util.module_from_spec(spec); spec.loader.exec_module(S)


def frame_fixture(n, stream, frame):
    c = np.concatenate([S.make_cloud(n, stream, frame), S.edge_case_points()])
    g = O.ref_ground_remove(c)
    pol = O.ref_ground_polar(c)
    cl = O.ref_cluster(g["elevated"])
    bx = O.ref_box_fit(g["elevated"], cl["grid"], cl["num_cluster"])
    # masks instead of clouds (outputs are order-preserving subsets of the input)
    return dict(cloud=c, n_elevated=len(g["elevated"]), n_ground=len(g["ground"]),
                elevated_head=g["elevated"][:64, :3], ground_head=g["ground"][:64, :3],
                elevated_xyz_sum=g["elevated"][:, :3].astype(np.float64).sum(0), ground_xyz_sum=g["ground"][:, :3].astype(np.float64).sum(0),
                min_z=pol["min_z"], height=pol["height"], is_ground=pol["is_ground"], hground=pol["hground"],
                grid=cl["grid"].astype(np.int16), num_cluster=cl["num_cluster"], boxes=bx["boxes"])


def frame0_fixture(n, stream, frame):
    """one frame through the SECOND package's own sources (object_tracking0: KITTI constants, oracle/_ref/libmot_ref0.so)"""
    c = np.concatenate([S.make_cloud(n, stream, frame), S.edge_case_points()])
    r = O.ref0_frame(c)
    return dict(cloud=c, n_elevated=len(r["elevated"]), n_ground=len(r["ground"]), elevated_head=r["elevated"][:64, :3],
                ground_head=r["ground"][:64, :3], elevated_xyz_sum=r["elevated"][:, :3].astype(np.float64).sum(0),
                ground_xyz_sum=r["ground"][:, :3].astype(np.float64).sum(0), grid=r["grid"].astype(np.int16),
                num_cluster=r["num_cluster"], boxes=r["boxes"])


def side_fixture(n, stream, frame):
    """cluster-node side products (makeClusteredCloud / setObsMsg / createCostMap) of one small frame"""
    elev = np.concatenate([O.ref_ground_remove(S.make_cloud(n, stream, frame))["elevated"], S.edge_case_points()])
    cl = O.ref_cluster(elev)
    r = O.ref_cluster_products(elev, cl["grid"])
    return dict(elevated=elev, grid=cl["grid"].astype(np.int16), clustered=r["clustered"], obstacles=r["obstacles"], cost_map=r["cost_map"])


def markers_fixture(n, stream, frame):
    """the rviz CUBE markers boxFitting fills (mark_cluster, box_fitting.cpp:161-209) for one frame: pose.position and scale per kept box"""
    elev = O.ref_ground_remove(S.make_cloud(n, stream, frame))["elevated"]
    cl = O.ref_cluster(elev)
    bx = O.ref_box_fit(elev, cl["grid"], cl["num_cluster"])
    return dict(elevated=elev, grid=cl["grid"].astype(np.int16), num_cluster=np.int32(cl["num_cluster"]), boxes=bx["boxes"],
                markers=O.ref_box_markers(elev, cl["grid"], cl["num_cluster"]))


def tracker_fixture(stream, nframes, npts, unit, ot0_workdir=None):
    """ot0_workdir: run object_tracking0's tracker instead (KITTI constants; it reads the ego motion from text files that
    Ref0Tracker writes under that directory); the boxes still come from the first package's chain (more of them)."""
    vs, yaws = 2.0 + 0.05 * np.arange(nframes), 0.004 * np.arange(nframes)
    if ot0_workdir is not None:   # the package's own fixtures: ego speed / yaw of KITTI 2011_09_26_drive_0005, one value per frame
        vs = np.loadtxt("/root/reference/object_tracking0/src/ego_velo.txt")[:nframes]
        yaws = np.loadtxt("/root/reference/object_tracking0/src/ego_yaw.txt")[:nframes]
    if ot0_workdir is None:
        R = O.RefTracker(); R.reset()
    else:
        R = O.Ref0Tracker(); R.reset(ot0_workdir, vs, yaws)
    boxes, n_boxes, tm, st, vis, pos, vyaw, ego = [], [], [], [], [], [], [], []
    states = []
    for f in range(nframes):
        c = S.make_cloud(npts, stream, f)
        g = O.ref_ground_remove(c); cl = O.ref_cluster(g["elevated"]); bx = O.ref_box_fit(g["elevated"], cl["grid"], cl["num_cluster"])
        b = bx["boxes"]
        ts = 1.0e9 + f * unit
        ego.append(R.ego_update(ts, vs[f], yaws[f]) if ot0_workdir is None else R.ego_update(ts))
        r = R.step(b, ts)
        pad = np.zeros((32, 8, 3), np.float32); pad[: len(b)] = b
        boxes.append(pad); n_boxes.append(len(b))
        T = 128
        def padv(a, shape, dt):
            o = np.zeros((T,) + shape, dt); o[: len(a)] = a; return o
        tm.append(padv(r["track_manage"], (), np.int32)); st.append(padv(r["is_static"], (), np.int32)); vis.append(padv(r["is_vis"], (), np.int32))
        pos.append(padv(r["p"], (3,), np.float32)); vyaw.append(padv(r["v_yaw"], (2,), np.float64))
        xs = np.zeros((T, 5)); ps = np.zeros((T, 25)); mp = np.zeros((T, 3)); lt = np.zeros(T, np.int32)
        for i in range(r["n"]):
            s = R.state(i); xs[i] = s["x_merge"]; ps[i] = s["p_merge"]; mp[i] = s["mode_prob"]; lt[i] = s["lifetime"]
        states.append((xs, ps, mp, lt, r["n"]))
    if ot0_workdir is not None:
        R.close()
    return dict(boxes=np.stack(boxes), n_boxes=np.array(n_boxes, np.int32), track_manage=np.stack(tm), is_static=np.stack(st),
                is_vis=np.stack(vis), pos=np.stack(pos), v_yaw=np.stack(vyaw), ego=np.stack(ego),
                x_merge=np.stack([s[0] for s in states]), p_merge=np.stack([s[1] for s in states]),
                mode_prob=np.stack([s[2] for s in states]), lifetime=np.stack([s[3] for s in states]),
                n_tracks=np.array([s[4] for s in states], np.int32), unit=unit, ego_v=vs, ego_yaw=yaws)


if __name__ == "__main__":
    assert O.ref() is not None, "oracle/_ref is not built (needs /root/reference)"
    fx = frame_fixture(9000, 3, 0)
    np.savez_compressed(os.path.join(HERE, "frame_ot_9k.npz"), **fx)
    fx = frame_fixture(24000, 5, 2)
    np.savez_compressed(os.path.join(HERE, "frame_ot_24k.npz"), **fx)
    np.savez_compressed(os.path.join(HERE, "side_ot_9k.npz"), **side_fixture(9000, 3, 0))
    np.savez_compressed(os.path.join(HERE, "markers_ot_24k.npz"), **markers_fixture(24000, 5, 2))
    assert O.ref0() is not None, "oracle/_ref/libmot_ref0.so is not built"
    np.savez_compressed(os.path.join(HERE, "frame_ot0_60k.npz"), **frame0_fixture(60000, 4, 1))
    for unit, name in ((1e5, "us"), (0.1, "sec")):
        np.savez_compressed(os.path.join(HERE, f"tracker_ot_{name}.npz"), **tracker_fixture(1, 30, 40000, unit))
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        np.savez_compressed(os.path.join(HERE, "tracker_ot0_us.npz"), **tracker_fixture(3, 30, 40000, 1e5, ot0_workdir=d))
    # the tracking node's sensor -> global change of frame (OT/tracking/main.cpp:76-83,143-158) on the tf / pcl_ros code of the
    # node-level oracle (oracle/ref_tf_capi.cpp): poses, boxes and their images, for tests/test_tf_exact.py
    import ctypes as C
    rng = np.random.default_rng(77)
    poses, boxes, glob = [], [], []
    for k in range(200):
        yaw = rng.uniform(-7, 7) if k % 5 else [0.0, np.pi, -np.pi, np.pi / 2, -np.pi / 2][(k // 5) % 5]
        pose = np.array([rng.uniform(-300, 300), rng.uniform(-300, 300), yaw])
        b = rng.uniform(-60, 60, size=(3, 8, 3)).astype(np.float32); out = np.zeros_like(b)
        assert O.ref().ref_boxes_to_global(b.ctypes.data_as(C.c_void_p), len(b), C.c_double(pose[0]), C.c_double(pose[1]), C.c_double(pose[2]), out.ctypes.data_as(C.c_void_p)) == 0
        poses.append(pose); boxes.append(b); glob.append(out)
    np.savez_compressed(os.path.join(HERE, "tf_boxes.npz"), pose=np.array(poses), boxes=np.array(boxes), **{"global": np.array(glob)})
    # the reference's only data fixture: the ego motion of KITTI drive_0005 its second package reads frame by frame
    # (OT0/src/imm_ukf_jpda.cpp:65-72). bench.py drives its synthetic 154-frame sequences with it.
    np.savez_compressed(os.path.join(HERE, "ego_drive0005.npz"),
                        ego_v=np.loadtxt("/root/reference/object_tracking0/src/ego_velo.txt"),
                        ego_yaw=np.loadtxt("/root/reference/object_tracking0/src/ego_yaw.txt"))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
