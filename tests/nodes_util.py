"""helpers shared by tests/test_nodes.py (CPU, emulator build of the kernels) and tests/test_nodes_gpu.py (libmot_hip.so):
input logs for the three nodes, running a node over a log, and the two comparisons (byte-equal / tracker tolerance)"""
import numpy as np

import nodes_build as NB
import roslog as R

T0, NF = 1.0e3, 6


def scans(synth, n=26000):
    """velodyne_points messages: a 32-byte point layout with the fields out of order on odd frames, NaN points included"""
    out = []
    for f in range(NF):
        c = np.concatenate([synth.make_cloud(n, 2, f), synth.edge_case_points()]).astype(np.float32)
        t = T0 + 0.1 * f
        if f % 2 == 0:
            out.append(R.pointcloud2(c, t, seq=f))
        else:   # x, y, z at offsets 4, 8, 16 inside 32-byte records
            m = R.pointcloud2(c[:, :1], t, seq=f, point_step=32)
            raw = np.zeros((len(c), 32), np.uint8)
            for k, off in enumerate((4, 8, 16, 24)):
                raw[:, off:off + 4] = c[:, k].copy().view(np.uint8).reshape(-1, 4)
            m["data"] = raw.reshape(-1)
            m["fields"] = [dict(name=nm, offset=off, datatype=7, count=1) for nm, off in (("intensity", 24), ("y", 8), ("x", 4), ("z", 16))]
            out.append(m)
    return out


def ground_log(synth):
    recs = []
    for f, m in enumerate(scans(synth)):
        recs += [("__now__", T0 + 0.1 * f + 0.01), ("velodyne_points", "sensor_msgs/PointCloud2", m)]
    return recs


def relay(out, topic, dt):
    recs = []
    for t, ty, b in out:
        if t == topic:
            recs += [("__now__", T0 + 0.1 * (len(recs) // 2) + dt), (t, ty, b)]
    return recs


def run(exe, recs, tmp, name, params=None, cwd=None):
    i, o = str(tmp / (name + "_in.log")), str(tmp / (name + "_out.log"))
    R.write_log(i, recs)
    NB.run_node(exe, i, o, params, cwd=cwd)
    return R.read_log(o)


def same(a, b):
    assert [(t, ty) for t, ty, _ in a] == [(t, ty) for t, ty, _ in b]
    for k, ((t, ty, x), (_, _, y)) in enumerate(zip(a, b)):
        if x != y:
            dx, dy = R.decode(ty, x), R.decode(ty, y)
            diff = [f for f in dx if repr(dx[f]) != repr(dy[f])]
            raise AssertionError(f"record {k} on {t}: fields that differ: {diff}")


def tracking_log(cluster_out):
    """track_box messages of a cluster node, re-stamped with the scan times (the chain itself publishes stamp 0: the ground
    node's output header only carries the frame id), interleaved with /gps/odom"""
    recs, f = [], 0
    for t, ty, b in cluster_out:
        if t != "track_box":
            continue
        m = R.decode(ty, b); m["header"]["stamp"] = R.stamp(T0 + 0.1 * f); m["header"]["seq"] = f
        od = dict(header=dict(seq=f, stamp=R.stamp(T0 + 0.1 * f), frame_id="gps"), child_frame_id="base_link",
                  pose=dict(pose=dict(orientation=dict(x=0.0, y=0.0, z=0.3 - 0.004 * f, w=1.0))),
                  twist=dict(twist=dict(linear=dict(x=3.0 + 0.1 * f, y=0.4, z=0.0))))
        recs += [("__now__", T0 + 0.1 * f + 0.03), ("/gps/odom", "nav_msgs/Odometry", od), ("track_box", ty, m)]
        f += 1
    return recs


def markers_close(a, b, rtol=1e-4):
    """tracker outputs are floating point (bar: 1e-4 relative); everything structural must be equal"""
    assert [(t, ty) for t, ty, _ in a] == [(t, ty) for t, ty, _ in b]
    for (t, ty, x), (_, _, y) in zip(a, b):
        mx, my = R.decode(ty, x), R.decode(ty, y)
        for k in ("header", "ns", "id", "type", "action", "color", "lifetime", "frame_locked", "text"):
            assert mx[k] == my[k], (t, k, mx[k], my[k])
        assert len(mx["points"]) == len(my["points"])
        vx = [mx["pose"]["position"][c] for c in "xyz"] + [mx["pose"]["orientation"][c] for c in "xyzw"] + [mx["scale"][c] for c in "xyz"]
        vy = [my["pose"]["position"][c] for c in "xyz"] + [my["pose"]["orientation"][c] for c in "xyzw"] + [my["scale"][c] for c in "xyz"]
        assert np.allclose(vx, vy, rtol=rtol, atol=1e-5), (mx["ns"], mx["id"], vx, vy)
        px = np.array([[p[c] for c in "xyz"] for p in mx["points"]]).reshape(-1, 3)
        py = np.array([[p[c] for c in "xyz"] for p in my["points"]]).reshape(-1, 3)
        assert np.allclose(px, py, rtol=rtol, atol=1e-5)



def reference_chain(ref_nodes, synth, tmp):
    """the reference's own nodes over the test sequence: ground -> cluster -> tracking (as chained: stamp 0; and re-stamped)"""
    g = run(ref_nodes["ground"], ground_log(synth), tmp, "ground")
    c = run(ref_nodes["cluster"], relay(g, "none_ground_topic", 0.02), tmp, "cluster")
    t0 = run(ref_nodes["tracking"], relay(c, "track_box", 0.03), tmp, "tracking_chain")        # stamp 0, no odometry
    t1 = run(ref_nodes["tracking"], tracking_log(c), tmp, "tracking_stamped")
    assert sum(t == "track_box" for t, _, _ in c) == NF and len(t1) >= 4 * NF
    assert any(R.decode(ty, b)["box_num"] > 0 for t, ty, b in c if t == "track_box")
    return dict(ground=g, cluster=c, tracking_chain=t0, tracking_stamped=t1)


def check_against_reference(nodes, ref, synth, tmp):
    g = run(nodes["ground"], ground_log(synth), tmp, "ground")
    same(ref["ground"], g)                                   # aux_points, none_ground_topic, ground_topic
    c = run(nodes["cluster"], relay(ref["ground"], "none_ground_topic", 0.02), tmp, "cluster")
    same(ref["cluster"], c)                                  # realtime_cost_map, cluster_obs, output, track_box, cluster_ma, visualization_marker
    t0 = run(nodes["tracking"], relay(ref["cluster"], "track_box", 0.03), tmp, "tracking_chain")
    markers_close(ref["tracking_chain"], t0)
    t1 = run(nodes["tracking"], tracking_log(ref["cluster"]), tmp, "tracking_stamped")
    markers_close(ref["tracking_stamped"], t1)


# ---------------------------------------------------------------- the single-process node of object_tracking0
NF0 = 14


def pipeline_setup(synth, tmp, n=60000):
    """input log (`input` topic) and the working directory with the two ego-motion text files OT0 reads relative to it"""
    import os
    d = tmp / "src" / "object_tracking" / "src"
    os.makedirs(d, exist_ok=True)
    # the first values of the package's own fixtures OT0/src/ego_velo.txt / ego_yaw.txt (KITTI drive_0005)
    velo = (3.51477, 3.48864, 3.46854, 3.40843, 3.34564, 3.30648, 3.28052, 3.23173, 3.20755, 3.18877, 3.16731, 3.16198, 3.14918, 3.12115)
    yaw = (-1.22191, -1.20608, -1.19362, -1.18118, -1.16722, -1.1547, -1.14217, -1.1288, -1.11721, -1.10575, -1.09323, -1.08131, -1.07012, -1.05775)
    assert len(velo) == NF0 and len(yaw) == NF0
    open(d / "ego_velo.txt", "w").write("".join("%.17g\n" % v for v in velo))
    open(d / "ego_yaw.txt", "w").write("".join("%.17g\n" % v for v in yaw))
    recs = []
    for f in range(NF0):
        c = np.concatenate([synth.make_cloud(n, 4, f), synth.edge_case_points()]).astype(np.float32)
        recs += [("__now__", T0 + 0.1 * f + 0.01), ("input", "sensor_msgs/PointCloud2", R.pointcloud2(c, T0 + 0.1 * f, frame_id="velo_link", seq=f))]
    return recs


def check_pipeline(ref_exe, own_exe, synth, tmp):
    import os
    recs = pipeline_setup(synth, tmp)
    os.makedirs(tmp / "a", exist_ok=True); os.makedirs(tmp / "b", exist_ok=True)
    a = run(ref_exe, recs, tmp / "a", "pipeline", cwd=str(tmp))
    b = run(own_exe, recs, tmp / "b", "pipeline", {"ego": "files"}, cwd=str(tmp))
    same([r for r in a if r[0] == "output"], [r for r in b if r[0] == "output"])
    ma, mb = [r for r in a if r[0] != "output"], [r for r in b if r[0] != "output"]
    markers_close(ma, mb)
    ns = [R.decode(ty, x)["ns"] for _, ty, x in ma]
    assert ns.count("boxes") == NF0 and ns.count("points") == 4 * NF0
    return ma
