"""a stage-wise call after a fused batch (shared by the emulator test and the GPU test)"""
import numpy as np
import pytest


def run(mot, lib, synth, oracle, upload=None, N=5000):
    # upload(host array) -> pointer frames_dev takes (the emulator reads host memory: the array itself)
    """The fused path leaves 12-byte elevated points (round 5); a stage-wise call puts float4 records into slot 0 only. Readers must take each
    slot's cloud for what it is: the on-demand labels, side products and cubes of the batch's OTHER slots after such a call, slot 0's own
    results, and mot_cluster_node_frame on a context that ran a fused batch before (its side products looked at the layout flag before the
    upload)."""
    B = 3; stride = ((N + 1023) // 1024) * 1024
    p = oracle.params(0)
    host = np.zeros((B, stride, 4), np.float32)
    for s in range(B):
        host[s, :N] = synth.make_cloud(N, 90 + s, s % 3)
    o = [oracle.ground_remove(p, host[s, :N]) for s in range(B)]
    want = [oracle.cluster(p, o[s]["elevated"]) for s in range(B)]
    other = np.ascontiguousarray(o[1]["elevated"][::-1])   # the stage-wise call's cloud: slot 1's, backwards
    ref_other = oracle.cluster(p, other)
    ptr = upload(host) if upload else host.ctypes.data
    for call in ("cluster", "box_fit", "products_host", "node_frame", "ground_remove", "smaller_fused_batch"):
        with mot.Context(**({"lib_path": lib} if lib else {}), max_points=stride, max_batch=B, max_tracks_total=64) as c:
            c.frames_dev(ptr, stride * 4, [N] * B)
            before = {s: (c.get_boxes(s)["boxes"], c.cluster_products(s), c.box_markers(s)) for s in (1, 2)}   # the batch's own answers, packed clouds
            if call == "cluster":
                got = c.cluster(other)
                assert np.array_equal(got["grid"], ref_other["grid"]) and np.array_equal(got["point_label"], ref_other["point_label"])
            elif call == "box_fit":
                got = c.box_fit(other, ref_other["grid"], ref_other["num_cluster"])
                assert np.array_equal(got["boxes"], oracle.box_fit(p, other, ref_other["grid"], ref_other["num_cluster"])["boxes"])
            elif call == "products_host":
                got = c.cluster_products_host(other, ref_other["grid"])
                ref = oracle.cluster_products(p, other, ref_other["grid"])
                assert np.array_equal(got["clustered"], ref["clustered"]) and np.array_equal(got["cost_map"], ref["cost_map"])
            elif call == "node_frame":
                got = c.cluster_node_frame(other)
                ref = oracle.cluster_products(p, other, ref_other["grid"])
                assert np.array_equal(got["clustered"], ref["clustered"]) and np.array_equal(got["obstacles"], ref["obstacles"]) and np.array_equal(got["cost_map"], ref["cost_map"])
                assert np.array_equal(got["boxes"], oracle.box_fit(p, other, ref_other["grid"], ref_other["num_cluster"])["boxes"])
            elif call == "ground_remove":
                got = c.ground_remove(host[0, :N])
                assert np.array_equal(got["elevated"], o[0]["elevated"])
            else:   # a fused call over slot 0 alone, with the ground cloud asked for (so it leaves float4 records): slots 1, 2 keep the first batch's 12-byte points
                c.set_fused_outputs(mot.OUT_GROUND)
                c.frames_dev(ptr, stride * 4, [N])
                got = c.get_ground(0, n_hint=N)
                assert np.array_equal(got["elevated"], o[0]["elevated"]) and np.array_equal(got["ground"], o[0]["ground"])
                assert np.array_equal(c.get_clusters(0, n_elevated=len(o[0]["elevated"]))["point_label"][: len(o[0]["elevated"])], want[0]["point_label"])
            for s in (1, 2):   # the other slots still hold the fused batch's 12-byte points
                cl = c.get_clusters(s, n_elevated=len(o[s]["elevated"]))
                assert np.array_equal(cl["point_label"][: len(o[s]["elevated"])], want[s]["point_label"]), (call, s)
                sp, mk = c.cluster_products(s), c.box_markers(s)
                assert all(np.array_equal(sp[k], before[s][1][k]) for k in ("clustered", "obstacles", "cost_map")), (call, s)
                assert np.array_equal(mk, before[s][2]) and np.array_equal(c.get_boxes(s)["boxes"], before[s][0]), (call, s)
                with pytest.raises(mot.MotError) as e:   # their float4 records cannot be rebuilt any more (slot 0 of the batch is gone): an error, not 12-byte points read as 16
                    c.get_ground(s, n_hint=N)
                assert e.value.code == mot.MOT_E_STATE, (call, s)
                # ... nor their ground cloud alone: after a stage-wise mot_ground_remove only slot 0's is resident
                gb = np.zeros((N, 4), np.float32); ne, ng = mot.C.c_int(0), mot.C.c_int(0)
                rc = c.lib.mot_get_ground(c._h, s, None, mot.C.byref(ne), gb.ctypes.data_as(mot.C.c_void_p), mot.C.byref(ng), None, N)
                assert rc == mot.MOT_E_STATE, (call, s, rc)


def run_ground_after_takeover(mot, lib, synth, oracle, upload=None, N=5000):
    """round-5 advice: after a fused batch that WROTE the ground cloud and the mask (mot_set_fused_outputs GROUND | MASK), a stage-wise cluster / box call
    replaces slot 0's elevated cloud — mot_get_ground(0) must then answer MOT_E_STATE (it used to return the new elevated cloud next to the old batch's
    ground cloud and mask); slot 1 keeps its own, complete ground result; a new ground stage on slot 0 makes it readable again."""
    B = 2; stride = ((N + 1023) // 1024) * 1024
    p = oracle.params(0)
    host = np.zeros((B, stride, 4), np.float32)
    for s in range(B):
        host[s, :N] = synth.make_cloud(N, 70 + s, s)
    o = [oracle.ground_remove(p, host[s, :N]) for s in range(B)]
    other = np.ascontiguousarray(o[1]["elevated"][::-1])
    ref_other = oracle.cluster(p, other)
    ptr = upload(host) if upload else host.ctypes.data
    for call in ("cluster", "box_fit", "products_host", "node_frame"):
        with mot.Context(**({"lib_path": lib} if lib else {}), max_points=stride, max_batch=B, max_tracks_total=64) as c:
            c.set_fused_outputs(mot.OUT_GROUND | mot.OUT_MASK)
            c.frames_dev(ptr, stride * 4, [N] * B)
            g0 = c.get_ground(0, n_hint=N)
            assert np.array_equal(g0["ground"], o[0]["ground"]) and np.array_equal(g0["mask"], o[0]["mask"])
            if call == "cluster":
                c.cluster(other)
            elif call == "box_fit":
                c.box_fit(other, ref_other["grid"], ref_other["num_cluster"])
            elif call == "products_host":
                c.cluster_products_host(other, ref_other["grid"])
            else:
                c.cluster_node_frame(other)
            with pytest.raises(mot.MotError) as e:
                c.get_ground(0, n_hint=N)
            assert e.value.code == mot.MOT_E_STATE, call
            g1 = c.get_ground(1, n_hint=N)   # the other slot of the batch: float4 clouds and mask were written by the batch itself, nothing to rebuild
            assert np.array_equal(g1["elevated"], o[1]["elevated"]) and np.array_equal(g1["ground"], o[1]["ground"]) and np.array_equal(g1["mask"], o[1]["mask"]), call
            c.frames_dev(ptr, stride * 4, [N] * B)   # a new ground stage: slot 0 is the batch's again
            g0 = c.get_ground(0, n_hint=N)
            assert np.array_equal(g0["elevated"], o[0]["elevated"]) and np.array_equal(g0["ground"], o[0]["ground"]) and np.array_equal(g0["mask"], o[0]["mask"]), call
