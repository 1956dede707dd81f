"""Stage 0 + Stage 1 of "stop reading the cloud twice"  on the MI355X — EXPERIMENT, not product.

  Stage 0  on 512 rendered bench frames (the product's own masks): the fraction of 64 / 128 / 256 / 1024-byte lines of the input cloud
           that hold at least one elevated point = what a re-load of the elevated points only has to fetch. Kill if > 0.7.
  Stage 1  tools/probes/frame_team_probe.hip (persistent frame-team kernel: hold {cell, z} in registers, stand-in filter, re-load the
           elevated points, write the elevated cloud + cells + occupancy) against the product's min-z + filter + compaction on the same
           512 frames in the same session; its elevated clouds are compared bit for bit with the product's. Kill if < 15 % better.

    python tools/probes/frame_team_probe.py [--frames 512] [--points 120000] > gpurun_out/<session>/frame_team_probe.txt
"""
import argparse
import ctypes as C
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--points", type=int, default=120000)
    ap.add_argument("--scenes", type=int, default=32)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--verify", type=int, default=24, help="frames whose elevated cloud is compared with the product's")
    a = ap.parse_args()
    import torch
    mot = _load("mot_amd", os.path.join(PKG, "__init__.py"))
    sdev = _load("mot_amd.synth_dev", os.path.join(os.path.dirname(PKG), "tools", "synth", "synth_dev.py"))
    dev = "cuda:0"
    B, N, S = a.frames, a.points, a.scenes
    F = B // S
    stride = ((N + 2047) // 2048) * 2048
    t0 = time.time()
    ego_v, ego_yaw = sdev.load_ego(154)
    # S scenes x F frames spread evenly over the bench's 154-frame drive
    seq_all, n_all, _, _ = sdev.SequenceRenderer(dev).render(list(range(S)), 154, N, stride, ego_v, ego_yaw)
    pick = np.linspace(0, 153, F).astype(int)
    frames = seq_all[torch.as_tensor(pick, device=dev)].reshape(B, stride, 4).contiguous()
    n = np.ascontiguousarray(n_all[pick].reshape(B), np.int32)
    del seq_all
    torch.cuda.empty_cache()
    print(f"rendered {B} frames ({S} scenes x {F} frames of the drive) of {N} points in {time.time() - t0:.1f} s; stride {stride}", flush=True)

    ctx = mot.Context(device=0, max_points=stride, max_batch=B, max_tracks_total=64)
    lib = ctx.lib
    ctx.set_fused_outputs(3)   # ground cloud + mask, for stage 0 and the comparison
    ctx.frames_dev(frames.data_ptr(), stride * 4, n)
    ctx.synchronize()
    hg = np.zeros((B, 9600), np.float32)
    masks, elev_ref, ne_ref = [], {}, np.zeros(B, np.int64)
    for b in range(B):
        assert lib.mot_debug_copy(ctx._h, 10, b, hg[b].ctypes.data_as(C.c_void_p), C.c_size_t(hg[b].nbytes)) == 0
        g = ctx.get_ground(b, int(n[b]))
        masks.append(g["mask"]); ne_ref[b] = g["n_elevated"]
        if b < a.verify:
            elev_ref[b] = g["elevated"]
    # ---------------------------------------------------------------- stage 0
    ELEV = 2   # MOT_MASK_ELEVATED (include/mot.h)
    assert int((masks[0] == ELEV).sum()) == ne_ref[0], "mask code"
    res = {"frames": B, "points": N, "elevated_fraction": float(ne_ref.sum() / n.sum())}
    for line_pts in (4, 8, 16, 64):
        tot = hit = 0
        for b in range(B):
            e = masks[b] == ELEV
            m = len(e) // line_pts * line_pts
            hit += int(e[:m].reshape(-1, line_pts).any(1).sum()); tot += m // line_pts
        res[f"lines_{16 * line_pts}B_with_an_elevated_point"] = hit / tot
    print("STAGE 0", json.dumps(res), flush=True)

    # ---------------------------------------------------------------- baseline: the product's ground stage on the same frames
    ctx.set_fused_outputs(0)
    ctx.frames_dev(frames.data_ptr(), stride * 4, n)
    ctx.synchronize()
    base = {}
    for name, sid in (("ground_stage", 0), ("polar_minz", 10), ("polar_filter", 11), ("classify_compact_elevated", 12)):
        ctx.time_stage(sid, B, 2)
        base[name] = ctx.time_stage(sid, B, a.iters) * 1e3
    print("BASELINE us per %d frames: " % B + json.dumps({k: round(v, 1) for k, v in base.items()}), flush=True)

    # ---------------------------------------------------------------- stage 1
    from tools_probes_build import build  # noqa: E402  (set up below)
    P = C.CDLL(build())

    class DevParams(C.Structure):
        _fields_ = [("raw", C.c_char * 512)]
    dp = DevParams()
    assert P.team_probe_params(C.byref(dp)) == 0
    P.team_probe_run.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p]
    d_n = torch.from_numpy(n).to(dev)
    d_hg = torch.from_numpy(hg).to(dev)
    out_e = torch.zeros((B, stride, 4), dtype=torch.float32, device=dev)
    ecell = torch.zeros((B, stride), dtype=torch.int16, device=dev)
    counts = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    Wmax = 16
    occ = torch.zeros((B, Wmax, 2, 2048), dtype=torch.int32, device=dev)
    team_minz = torch.zeros((256, 9600), dtype=torch.int32, device=dev)
    team_hg = torch.zeros((256, 9600), dtype=torch.float32, device=dev)
    merged = torch.zeros((B, 9600), dtype=torch.int32, device=dev)
    sync = torch.zeros((256, 32), dtype=torch.int32, device=dev)
    ticket = torch.zeros(4, dtype=torch.int32, device=dev)
    stamps = torch.zeros((1024, 8), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    rows = []
    for tb, pts in ((512, 48), (1024, 24), (1024, 16)):
        W = -(-int(n.max()) // (tb * pts))
        for filter_us, reload in ((18.0, 1), (0.0, 1), (35.0, 1), (18.0, 0)):
            counts.zero_(); out_e.zero_(); torch.cuda.synchronize()
            ms = C.c_float(0); resident = C.c_int(0)
            rc = P.team_probe_run(frames.data_ptr(), stride, d_n.data_ptr(), B, d_hg.data_ptr(), out_e.data_ptr(), stride, counts.data_ptr(), ecell.data_ptr(),
                                  occ.data_ptr(), team_minz.data_ptr(), team_hg.data_ptr(), merged.data_ptr(), sync.data_ptr(), ticket.data_ptr(), W, 4096, tb, pts,
                                  C.c_float(filter_us), reload, a.iters, C.byref(dp), None, C.byref(ms), C.byref(resident), stamps.data_ptr())
            torch.cuda.synchronize()
            stuck = int((sync[:, 2] != 0).sum())
            cnt = counts.cpu().numpy()
            # the counts accumulate the ground atomics over the 1 + iters launches; the elevated count is a plain store
            ok_counts = bool(np.array_equal(cnt[:, 0], ne_ref))
            ok_cloud = True
            if reload == 1 and filter_us == 18.0:
                oe = out_e[: a.verify].cpu().numpy()
                for b in range(a.verify):
                    if not np.array_equal(oe[b, : ne_ref[b]].view(np.uint32), elev_ref[b].view(np.uint32)):
                        ok_cloud = False
            st = stamps[: (min(resident.value, 4096) // W) * W].cpu().numpy().astype(np.float64)
            per_frame = (st[:, :7] / np.maximum(st[:, 7:8], 1)).mean(0) / 100.0   # us per frame and workgroup, mean over the workgroups
            row = dict(block=tb, points_per_thread=pts, team_wgs=W, resident_wgs=resident.value, teams=min(resident.value, 4096) // W, filter_us=filter_us, reload_only_elevated=bool(reload),
                       us_per_launch=round(ms.value * 1e3, 1), rc=rc, teams_gave_up=stuck, counts_equal_product=ok_counts, clouds_equal_product=ok_cloud,
                       vs_ground_stage=round(ms.value * 1e3 / base["ground_stage"], 3),
                       us_per_frame_in_phases=dict(zip(("load_cells_fold", "grid_out", "rendezvous_filter", "thresholds_in", "classify_scan", "reload_write", "occupancy_out"), [round(float(x), 1) for x in per_frame])))
            rows.append(row)
            print("STAGE 1", json.dumps(row), flush=True)
    best = min((r for r in rows if r["reload_only_elevated"] and r["filter_us"] == 18.0 and r["counts_equal_product"] and r["teams_gave_up"] == 0), key=lambda r: r["us_per_launch"], default=None)
    print("VERDICT", json.dumps(dict(ground_stage_us=round(base["ground_stage"], 1), best_probe=best,
                                     gain=None if best is None else round(1 - best["us_per_launch"] / base["ground_stage"], 3), go=bool(best and best["us_per_launch"] < 0.85 * base["ground_stage"]))))


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import build_probes as tools_probes_build
    sys.modules["tools_probes_build"] = tools_probes_build
    main()
