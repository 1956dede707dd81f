"""Node-level parity: what the three ROS nodes PUBLISH for the same input messages.

The reference's own node sources (OT/src/groundremove/main.cpp, OT/src/cluster/main.cpp, OT/tracking/main.cpp), main() included
and unmodified, are built against the file-backed mini-ROS of oracle/ref_shim into offline executables (oracle/_ref/bin) and
compared, topic by topic and byte by byte, with
  * "recipe": the same sources after the one edit INTEGRATION.md prescribes (algorithm include -> "mot_adapters.hpp"), linked
    against the C-ABI library, and
  * "own": this repository's node shells ros/src/*_node.cpp.
CPU only: the C-ABI library is the emulator build of the kernels (tests/emu), so this checks the host glue around the kernels
(message decode, pre-filters, message assembly, markers, tf), not the kernels themselves — those are the -m gpu tests.
ROS, tf and PCL are not installed here: the shim restates the slices the nodes use ("parity unpinned" for the shim itself)."""
import os
import sys

import numpy as np
import pytest

import nodes_build as NB
import roslog as R

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
pytestmark = pytest.mark.skipif(not NB.have_reference(), reason="the reference sources are not on this box")
T0, NF = 1.0e3, 6


@pytest.fixture(scope="module")
def emu_lib():
    import build_emu
    return build_emu.build()


@pytest.fixture(scope="module")
def ref_nodes():
    return NB.reference_nodes()


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


def run(exe, recs, tmp, name, params=None):
    i, o = str(tmp / (name + "_in.log")), str(tmp / (name + "_out.log"))
    R.write_log(i, recs)
    NB.run_node(exe, i, o, params)
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


@pytest.fixture(scope="module")
def reference_chain(ref_nodes, synth, tmp_path_factory):
    tmp = tmp_path_factory.mktemp("refchain")
    g = run(ref_nodes["ground"], ground_log(synth), tmp, "ground")
    c = run(ref_nodes["cluster"], relay(g, "none_ground_topic", 0.02), tmp, "cluster")
    t0 = run(ref_nodes["tracking"], relay(c, "track_box", 0.03), tmp, "tracking_chain")        # stamp 0, no odometry
    t1 = run(ref_nodes["tracking"], tracking_log(c), tmp, "tracking_stamped")
    assert sum(t == "track_box" for t, _, _ in c) == NF and len(t1) >= 4 * NF
    assert any(R.decode(ty, b)["box_num"] > 0 for t, ty, b in c if t == "track_box")
    return dict(ground=g, cluster=c, tracking_chain=t0, tracking_stamped=t1)


@pytest.mark.parametrize("kind", ["recipe", "own"])
def test_nodes_publish_what_the_reference_nodes_publish(kind, emu_lib, reference_chain, synth, tmp_path):
    nodes = NB.recipe_nodes(emu_lib) if kind == "recipe" else NB.own_nodes(emu_lib)
    ref = reference_chain
    g = run(nodes["ground"], ground_log(synth), tmp_path, "ground")
    same(ref["ground"], g)                                   # aux_points, none_ground_topic, ground_topic
    c = run(nodes["cluster"], relay(ref["ground"], "none_ground_topic", 0.02), tmp_path, "cluster")
    same(ref["cluster"], c)                                  # realtime_cost_map, cluster_obs, output, track_box, cluster_ma, visualization_marker
    t0 = run(nodes["tracking"], relay(ref["cluster"], "track_box", 0.03), tmp_path, "tracking_chain")
    markers_close(ref["tracking_chain"], t0)
    t1 = run(nodes["tracking"], tracking_log(ref["cluster"]), tmp_path, "tracking_stamped")
    markers_close(ref["tracking_stamped"], t1)


def test_ground_node_parameters(emu_lib, ref_nodes, synth, tmp_path):
    """filter_z_max / filter_z_min (OT/src/groundremove/main.cpp:147-148) reach the device pre-filter"""
    own = NB.own_nodes(emu_lib)
    prm = {"filter_z_max": 0.4, "filter_z_min": -1.9}
    recs = ground_log(synth)[:4]
    same(run(ref_nodes["ground"], recs, tmp_path, "ref", prm), run(own["ground"], recs, tmp_path, "own", prm))
