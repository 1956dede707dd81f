"""shared body of tests/test_emu_frame_kernel.py (emulator) and tests/test_frame_kernel_gpu.py (MI355X): the fused path with the
frame-per-workgroup compaction kernel forced on (mot_debug_option 0 = 1) against the oracle and against the chunk-per-workgroup
kernel — clouds, mask, counts, the cluster grid (i.e. the occupancy planes the kernel leaves), boxes, tracks."""
import ctypes as C

import numpy as np

CAND = np.dtype([("pc", "f4", 8), ("max_z", "f4"), ("accepted", "i4"), ("undefined", "i4"), ("branch", "i4"), ("poly_off", "i4"), ("poly_n", "i4"),
                 ("off_x", "i4"), ("off_y", "i4"), ("num_points", "i4"), ("pad", "i4")])   # BoxCandidate, csrc/mot_internal.h


def _dbg(ctx, which, slot, dtype, count):
    out = np.zeros(count, dtype)
    assert ctx.lib.mot_debug_copy(ctx._h, which, slot, out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes)) == 0
    return out


def many_clusters_cloud(seed=5):
    """isolated blobs on every fifth cell of the cluster grid, three returns each, lifted off the road: several hundred clusters"""
    import patterns
    rng = np.random.default_rng(seed)
    G, roi = 250, 50.0
    cells = [(x, y) for x in range(2, G - 2, 5) for y in range(2, G - 2, 5) if (x - G // 2) ** 2 + (y - G // 2) ** 2 > 18 ** 2]
    pts = patterns.cells_to_points(cells, G, roi, 3, rng)
    pts[:, 2] = rng.uniform(-0.9, 0.2, len(pts))
    return pts[rng.permutation(len(pts))]


def crowded_cloud(oracle, synth, n_base, copies, seed=3):
    """a scan whose obstacle returns are repeated `copies` times with millimetre jitter (same cells, same ground): the number
    of elevated points grows past the 65536 the frame kernel keeps labels for, the scene stays the same"""
    base = synth.make_cloud(n_base, seed, 0)
    e = oracle.ground_remove(oracle.params(0), base)["elevated"]
    rng = np.random.default_rng(seed)
    extra = [e + np.concatenate([rng.normal(0, 0.002, (len(e), 2)), np.zeros((len(e), 2))], axis=1).astype(np.float32) for _ in range(copies)]
    return np.concatenate([base] + extra).astype(np.float32)


def run(mot, oracle, synth, lib_path, sizes, stride, preset=0, frames=2, crop=False, clouds_override=None):
    p = oracle.params(preset)
    kw = {"lib_path": lib_path} if lib_path else {}
    B = len(sizes)
    seen = dict(clusters=0, boxes=0, lshape=0)
    over = dict(crop_enable=1, crop_x_min=-20.0, crop_x_max=30.0, crop_y_min=-15.0, crop_y_max=25.0, crop_z_min=-2.5, crop_z_max=1.0) if crop else {}
    with mot.Context(mot.params(preset, **over), max_points=stride, max_batch=B, max_tracks_total=128, **kw) as a, \
         mot.Context(mot.params(preset, **over), max_points=stride, max_batch=B, max_tracks_total=128, **kw) as b:
        assert a.lib.mot_debug_option(a._h, 0, 1) == 0     # a: one workgroup per frame
        assert b.lib.mot_debug_option(b._h, 0, 0) == 0     # b: one workgroup per chunk
        for f in range(frames):
            host = np.zeros((B, stride, 4), np.float32)
            clouds = []
            for s, n in enumerate(sizes):
                c = clouds_override[s][:n] if clouds_override else synth.make_cloud(max(n, 1), 30 + s, f)[:n]
                host[s, :n] = c
                clouds.append(c)
            args = dict(run_tracker=True, timestamps=[1.0e9 + f * 1e5] * B, ego_v=[1.5] * B, ego_yaw=[0.01 * f] * B)
            if lib_path:
                a.frames_dev(host.ctypes.data, stride * 4, sizes, **args); b.frames_dev(host.ctypes.data, stride * 4, sizes, **args)
            else:
                import hiprt
                dev = hiprt.DeviceBuffer(host)
                a.frames_dev(dev.ptr, stride * 4, sizes, **args); b.frames_dev(dev.ptr, stride * 4, sizes, **args)
                a.synchronize(); b.synchronize(); dev.free()
            for s, n in enumerate(sizes):
                ra, rb = a.get_ground(s, n_hint=n), b.get_ground(s, n_hint=n)
                for k in ("elevated", "ground"):
                    assert np.array_equal(ra[k], rb[k]), (f, s, k)
                assert np.array_equal(ra["mask"][:n], rb["mask"][:n])
                ca, cb = a.get_clusters(s, n_elevated=len(ra["elevated"])), b.get_clusters(s, n_elevated=len(rb["elevated"]))
                assert ca["num_cluster"] == cb["num_cluster"] and np.array_equal(ca["grid"], cb["grid"]) and np.array_equal(ca["point_label"], cb["point_label"])
                assert np.array_equal(a.get_boxes(s)["boxes"], b.get_boxes(s)["boxes"])
                # what the fused label + index kernel leaves for the per-cluster kernels, and what they made of it
                nc, ne = ca["num_cluster"], len(ra["elevated"])
                cst = [_dbg(x, 7, s, np.int32, nc + 1) for x in (a, b)]
                assert np.array_equal(cst[0], cst[1])
                srt = [_dbg(x, 5, s, np.int32, max(int(cst[0][nc]), 1))[: int(cst[0][nc])] for x in (a, b)]
                assert np.array_equal(srt[0], srt[1])
                assert np.array_equal(_dbg(a, 8, s, np.int32, max(ne, 1))[:ne], _dbg(b, 8, s, np.int32, max(ne, 1))[:ne])
                cand = [_dbg(x, 0, s, CAND, max(nc, 1))[:nc] for x in (a, b)]
                for k in ("pc", "max_z", "accepted", "undefined", "branch", "poly_n", "off_x", "off_y", "num_points"):   # (poly_off: pool slots are handed out in arrival order)
                    assert np.array_equal(cand[0][k], cand[1][k], equal_nan=(k in ("pc", "max_z"))), (f, s, k)
                seen["clusters"] += nc; seen["boxes"] += len(a.get_boxes(s)["boxes"]); seen["lshape"] += int((cand[0]["branch"] == 0).sum())
                ta, tb = a.get_tracks(s), b.get_tracks(s)
                assert ta["n"] == tb["n"] and np.array_equal(ta["track_manage"], tb["track_manage"]) and np.array_equal(ta["p"], tb["p"])
                if not crop:   # and the oracle (the crop is the node's pre-filter: compared through the chunk kernel's own oracle tests)
                    g = oracle.ground_remove(p, clouds[s])
                    assert np.array_equal(ra["elevated"], g["elevated"]) and np.array_equal(ra["ground"], g["ground"]) and np.array_equal(ra["mask"][:n], g["mask"])
                    o = oracle.cluster(p, g["elevated"])
                    assert ca["num_cluster"] == o["num_cluster"] and np.array_equal(ca["grid"], o["grid"])
    assert seen["clusters"] > 0 and (seen["boxes"] > 0 or max(sizes) < 8192), seen   # the comparison was not vacuous (thin clouds make clusters too small for boxes)
    return seen
