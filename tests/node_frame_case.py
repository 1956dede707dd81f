"""mot_cluster_node_frame / mot_ground_node_frame (one call per node callback) against the call-by-call entry points they bundle — which are
compared with the oracle / the reference build elsewhere (test_cluster_box_gpu.py, test_ground_gpu.py) — and, for the boxes and the clouds,
against the oracle directly. Shared by the emulator test and the -m gpu test."""
import numpy as np


def check(ctx, oracle, synth, sizes=((30000, 5, 0), (12000, 2, 3))):
    p = oracle.params(0)
    for n, stream, frame in sizes:
        cloud = synth.make_cloud(n, stream, frame)
        og = oracle.ground_remove(p, cloud)
        g = ctx.ground_node_frame(cloud)
        assert np.array_equal(g["elevated"].view(np.uint32), og["elevated"].view(np.uint32)) and np.array_equal(g["ground"].view(np.uint32), og["ground"].view(np.uint32))
        e = og["elevated"]
        # call by call: mot_cluster + mot_cluster_products + mot_box_fit_resident + mot_box_markers
        cl = ctx.cluster(e); sd = ctx.cluster_products(0); bx = ctx.box_fit_resident(); mk = ctx.box_markers(0)
        fr = ctx.cluster_node_frame(e)
        assert fr["num_cluster"] == cl["num_cluster"] and fr["n_undefined"] == bx["n_undefined"]
        assert np.array_equal(fr["clustered"].view(np.uint32), sd["clustered"].view(np.uint32)) and np.array_equal(fr["obstacles"].view(np.uint32), sd["obstacles"].view(np.uint32))
        assert np.array_equal(fr["cost_map"], sd["cost_map"])
        assert len(fr["boxes"]) > 0 and np.array_equal(fr["boxes"].view(np.uint32), bx["boxes"].view(np.uint32)) and np.array_equal(fr["box_cluster"], bx["box_cluster"])
        assert np.array_equal(fr["cubes"].view(np.uint32), mk.view(np.uint32))
        ob = oracle.box_fit(p, e, oracle.cluster(p, e)["grid"], cl["num_cluster"])
        assert np.array_equal(fr["boxes"].view(np.uint32), ob["boxes"].view(np.uint32))
        # the resident state is what the call-by-call entry points leave: the getters agree
        assert ctx.get_clusters(0, len(e))["num_cluster"] == cl["num_cluster"] and np.array_equal(ctx.box_markers(0).view(np.uint32), mk.view(np.uint32))
    # an empty cloud, and one without a cluster
    fr = ctx.cluster_node_frame(np.zeros((0, 4), np.float32))
    assert fr["num_cluster"] == 0 and len(fr["boxes"]) == 0 and len(fr["clustered"]) == 0 and len(fr["obstacles"]) == 0
    g = ctx.ground_node_frame(np.zeros((0, 4), np.float32))
    assert len(g["elevated"]) == 0 and len(g["ground"]) == 0
    one = np.array([[5.0, 5.0, 0.5, 0.0]], np.float32)
    fr = ctx.cluster_node_frame(one)
    assert fr["num_cluster"] == 0 and len(fr["boxes"]) == 0
