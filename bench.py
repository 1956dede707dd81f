#!/usr/bin/env python3
"""bench.py — frames/s of the LiDAR perception hot path on MI355X (metric of BASELINE.json).

One "step" = one pass of the whole hot path (ground removal -> grid clustering -> box fit -> IMM-UKF-PDA tracker
step) over one batch of synthetic frames: B independent 64-beam sensor streams ("slots"), one ~120k-point frame
per stream, inputs already resident in HBM when the timed region starts. Consecutive steps feed consecutive frames
of each stream (moving obstacles), so the trackers really run. N GPUs = N processes (torch.distributed over RCCL),
each with its own B streams (weak scaling); what crosses GPUs each step is the fixed-size block of live-track
records per stream, all-gathered over xGMI.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      dominant kernel: algorithmic HBM bytes per launch / mean HIP-event time, vs 8 TB/s
  cpu_baseline  the reference's own sources (oracle/_ref) — or the C restatement if that library is absent —
                timed on this box's host cores on a bounded sample (1 thread: the reference is single-threaded).
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd")
HBM_PEAK_GBS = 8000.0
TRACK_RECORD_BYTES = 144
GATHER_TRACKS = 64  # live tracks per stream in the all-gathered block (BASELINE.json configs[3]: <= 64 tracks)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def cpu_baseline(synth, n_points, budget_s=12.0):
    """reference CPU path on a bounded sample: ground -> cluster -> box -> tracker, single thread"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    try:
        use_ref = O.ref() is not None
    except Exception:
        use_ref = False
    p = O.params(0)
    frames = [synth.make_cloud(n_points, 900, f) for f in range(6)]
    trk = O.RefTracker() if use_ref else O.Tracker(p)
    trk.reset()
    state = {"f": 0}

    def one(c):
        f = state["f"]; state["f"] += 1
        if use_ref:
            g = O.ref_ground_remove(c)
            cl = O.ref_cluster(g["elevated"])
            bx = O.ref_box_fit(g["elevated"], cl["grid"], cl["num_cluster"])["boxes"]
        else:
            g = O.ground_remove(p, c)
            cl = O.cluster(p, g["elevated"])
            bx = O.box_fit(p, g["elevated"], cl["grid"], cl["num_cluster"])["boxes"]
        ts = 1.0e9 + f * 1e5
        trk.ego_update(ts, 0.0, 0.0)
        trk.step(bx, ts, max_tracks=65536)

    one(frames[0])  # warm-up
    t0 = time.perf_counter(); k = 0
    while True:
        one(frames[k % len(frames)]); k += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or k >= 300:
            break
    return {"value": round(k / dt, 2), "unit": "frames/s", "cores": 1, "kind": "reference" if use_ref else "port",
            "sample": f"{k} frames x {n_points} pts, ground+cluster+box+tracker, single thread "
                      f"({os.cpu_count()} host cores present; the reference is single-threaded)"}


def host_boundary(mot, synth, n_points, frames=40, lib_path=None, device=0):
    """PCIe-INCLUSIVE rate of the host-buffer boundary (never the headline value): one stream, every frame starts as a
    PointCloud2-style payload in pageable host memory, is uploaded (mot_frame_pointcloud2: unpack, ground removal, clustering,
    box fit on the resident copy), its boxes are read back and the tracker steps on them — the per-frame sequence of the
    single-process node (ros/src/pipeline_node.cpp) without the ROS glue. Wall clock per frame, synchronous."""
    clouds = [np.ascontiguousarray(synth.make_cloud(n_points, 950, f)) for f in range(4)]
    payload = [c.view(np.uint8).reshape(-1) for c in clouds]
    kw = dict(lib_path=lib_path) if lib_path else dict(device=device)
    with mot.Context(max_points=((n_points + 2047) // 2048) * 2048, max_batch=1, max_tracks_total=4096, **kw) as c:
        def one(f):
            c.frame_pointcloud2(payload[f % 4], n_points, 16, 0, 4, 8)
            bx = c.get_boxes(0)["boxes"]
            ts = 1.0e9 + f * 1.0e5
            c.ego_update(ts, 0.0, 0.0)
            return c.track_step(bx, ts)["n"]
        for f in range(3):
            one(f)
        t0 = time.perf_counter()
        for f in range(3, 3 + frames):
            one(f)
        dt = time.perf_counter() - t0
    return {"value": round(frames / dt, 1), "unit": "frames/s", "ms_per_frame": round(dt / frames * 1e3, 4),
            "what": f"1 stream, {frames} frames x {n_points} pts from pageable host memory: H2D of the raw records + ground/cluster/box + "
                    "D2H of the boxes + tracker step + D2H of the tracks, synchronous per frame (PCIe-inclusive; not the headline value)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="sensor streams (one frame each) per GPU per step")
    ap.add_argument("--points", type=int, default=120000)
    ap.add_argument("--frames", type=int, default=4, help="distinct consecutive frames resident per stream")
    ap.add_argument("--contexts", type=int, default=4, help="contexts (HIP streams) per GPU the streams are split over: the "
                    "latency-bound kernels of one (CCL, polygon, tracker: one workgroup per stream) overlap the streaming kernels of the other")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # one hardware queue per context stream (HIP's default is 4 queues per process, shared with its own null stream:
    # with 4 contexts two of them would share a queue and serialise — measured 319 k vs 402 k frames/s)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch  # torch first: it brings its own HIP runtime, which libmot_hip.so then shares
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the library has no CPU fallback)", file=sys.stderr); sys.exit(2)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    mot = _load("mot_amd", os.path.join(PKG_DIR, "__init__.py"))
    synth = _load("mot_amd.synth", os.path.join(PKG_DIR, "synth.py"))
    multi = _load("mot_amd.multi", os.path.join(PKG_DIR, "multi.py"))
    build = _load("mot_amd.build", os.path.join(PKG_DIR, "build.py"))
    if not os.path.exists(build.LIB):
        build.build()

    B, N, F = args.batch, args.points, args.frames
    stride = ((N + 2047) // 2048) * 2048
    # synthetic streams: 8 distinct scenes per rank tiled over the B slots, F consecutive frames of each
    n_scene = min(B, 8)
    scenes = [[synth.make_cloud(N, multi.scene_of(rank, s, n_scene), f) for f in range(F)] for s in range(n_scene)]
    dev_frames = []
    for f in range(F):
        host = np.zeros((B, stride, 4), np.float32)
        for b in range(B):
            host[b, :N] = scenes[b % n_scene][f]
        dev_frames.append(torch.from_numpy(host).cuda())
    NC = max(1, min(args.contexts, B))
    assert B % NC == 0, "--batch must be divisible by --contexts"
    Bc = B // NC
    ctxs = [mot.Context(device=local, max_points=stride, max_batch=Bc, max_tracks_total=4096) for _ in range(NC)]
    ctx = ctxs[0]
    gathers = [multi.TrackGather(Bc, GATHER_TRACKS, world, "cuda") for _ in range(NC)] if world > 1 else None
    torch.cuda.synchronize()

    step_no = [0]
    host_issue = [0.0]   # seconds the host spent inside the asynchronous launch calls (if this approaches the step time, the host bounds the pipeline)
    sizes = np.full(Bc, N, np.int32); zeros = np.zeros(Bc, np.float64)

    def step():
        k = step_no[0]; step_no[0] += 1
        ts = np.full(Bc, 1.0e9 + (k % 200) * 1.0e5, np.float64)  # microsecond stamps => dt = 0.1 s
        t_h = time.perf_counter()
        for ci, cx in enumerate(ctxs):  # asynchronous launches on NC HIP streams
            if k % 200 == 0:
                cx.reset()  # a stream restarts: the reference never frees tracks, so long runs are cut into sequences
            cx.frames_dev(dev_frames[k % F].data_ptr() + ci * Bc * stride * 16, stride * 4, sizes, run_tracker=True, timestamps=ts,
                          ego_v=zeros, ego_yaw=zeros)
        host_issue[0] += time.perf_counter() - t_h
        if world > 1:  # the per-step result blocks cross GPUs over RCCL / xGMI
            for ci, cx in enumerate(ctxs):
                gathers[ci].step(cx)

    def sync_all():
        for cx in ctxs:
            cx.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    host_issue[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        # ---- per-kernel timing on the resident data (HIP events on the context's stream, see mot_time_stage)
        it = 20
        tr0 = ctx.get_tracks(0)
        kernels = {"polar_minz_kernel": 10, "polar_filter_kernel": 11, "classify_compact_kernel": 12,
                   "cart_occupancy_kernel": 20, "ccl_kernel": 21, "label_stats_kernel": 30, "cluster_index_kernel": 34, "cluster_gather_kernel": 31,
                   "cluster_rect_kernel": 33, "box_finalize_kernel": 32, "track_step_kernel": 40}
        k_ms = {k: ctx.time_stage(v, Bc, it if v != 40 else 5) for k, v in kernels.items()}
        stage_ms = {"ground": ctx.time_stage(0, Bc, it), "cluster": ctx.time_stage(1, Bc, it), "box": ctx.time_stage(2, Bc, it),
                    "stateless": ctx.time_stage(100, Bc, it)}
        counts = [ctx.get_ground(b, want_clouds=False) for b in range(Bc)]
        ne_tot = sum(c["n_elevated"] for c in counts); ng_tot = sum(c["n_ground"] for c in counts)
        cl0 = ctx.get_clusters(0); bx0 = ctx.get_boxes(0)
        G = ctx.params.num_grid
        # algorithmic HBM bytes per launch (DESIGN.md §"bytes per unit"): what a kernel must read and write once
        BL = Bc  # frames per launch (one context)
        alg_bytes = {"polar_minz_kernel": 16.0 * N * BL,
                     "polar_filter_kernel": 8.0 * 9600 * BL,
                     "classify_compact_kernel": 16.0 * N * BL + 16.0 * (ne_tot + ng_tot) + 1.0 * N * BL,
                     "cart_occupancy_kernel": 16.0 * ne_tot,
                     "ccl_kernel": (2 * 2048 * 4 + 4.0 * G * G) * BL,
                     "label_stats_kernel": (16.0 + 4.0) * ne_tot,
                     "cluster_index_kernel": 4.0 * ne_tot,
                     "cluster_gather_kernel": (4.0 + 16.0) * ne_tot,
                     "cluster_rect_kernel": 4.0 * ne_tot / 8,
                     "box_finalize_kernel": 96.0 * BL,
                     "track_step_kernel": (2 * 1624.0 + 144.0) * max(tr0["n"], 1) * BL}
        # the dominant kernel of an HBM roofline is the one that moves the most bytes (it also has the largest share of GPU
        # time in the rocprofv3 trace of this command, profiles/); the longest single launch is reported next to it
        dom = max(alg_bytes, key=lambda k: alg_bytes[k])
        longest = max(k_ms, key=lambda k: k_ms[k])
        achieved = alg_bytes[dom] / (k_ms[dom] * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KB), same launch size only
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_B128.json")
        if BL == 128 and N == 120000 and os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path)).get(dom)
            if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                traffic = int((2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)
        frame_bytes = sum(alg_bytes.values()) / BL
        frames = B * args.steps * world
        out = {
            "metric": "LiDAR frames/sec (120k-pt 64-beam cloud) end-to-end ground->cluster->track",
            "value": round(frames / dt, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "host_issue_ms_per_step": round(host_issue[0] / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32/f64 (fp32 grid indices with fp64 intermediates, int32 grids/labels, fp64 tracker — the reference's types)",
            "data": "synthetic",
            "config": {"workload": "configs[3]: ground removal -> CCL -> box fit -> batched IMM-UKF-PDA tracker on one MI355X, "
                                   "120k-pt synthetic HDL-64E clouds, one frame per stream per step",
                       "points_per_frame": N, "frames_per_step_per_gpu": B, "streams": B * world,
                       "contexts_per_gpu": NC, "frames_per_launch": BL, "elevated_pts_per_frame": ne_tot // BL, "clusters_frame0": cl0["num_cluster"], "boxes_frame0": len(bx0["boxes"]),
                       "tracks_stream0": int(tr0["n"]), "live_tracks_stream0": int((tr0["track_manage"] > 0).sum()),
                       "parallelism": f"stream-sharded x{world}" + (", all_gather of live-track records (RCCL)" if world > 1 else "")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": "profiles/r01_pmc_B128.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command at 128 frames per launch)" if traffic else None,
                         "longest_launch": {"kernel": longest, "ms": round(k_ms[longest], 5)},
                         "kernel_ms": {k: round(v, 5) for k, v in k_ms.items()},
                         "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()},
                         "algorithmic_bytes_per_launch": {k: int(v) for k, v in alg_bytes.items()},
                         "pipeline_bytes_per_frame": int(frame_bytes),
                         "pipeline_frac": round(frame_bytes * B / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if world == 1:
            try:
                out["host_boundary"] = host_boundary(mot, synth, N, device=local)
            except Exception as e:   # an auxiliary figure must never cost the bench line
                out["host_boundary"] = None
                print(f"host_boundary measurement failed: {e}", file=sys.stderr)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(synth, N)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
