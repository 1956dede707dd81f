"""Synthetic lidar SEQUENCES rendered on the GPU — BENCH / TEST DATA GENERATOR (tools/synth/synth.hip -> tools/synth/libmot_synth.so), not part of
the product package: libmot_hip.so and include/mot.h do not depend on it.

BASELINE.json's sequence configuration is the 154-frame KITTI drive_0005; no KITTI data exists here, so a
street of oriented boxes — parked and moving cars, pedestrians, walls, poles — is laid out along the path the ego vehicle
drives when it is fed the reference's own ego-motion fixture (OT0/src/ego_velo.txt / ego_yaw.txt, committed as
tests/golden/ego_drive0005.npz by tests/golden/make_golden.py), and every frame is ray-cast in the sensor's pose of that
frame. The sensor's motion follows the tracker's dead reckoning exactly (getOriginPoints, OT/tracking/imm_ukf_jpda.cpp:
74-172: heading = yaw - yaw_0, step = dt * v along the new heading), so static objects stand still in the tracker's global
frame and moving ones move with constant velocity — the tracker sees what it would see on a real drive.

torch is plumbing here (device memory, a cumulative sum, a scatter); the rays are cast by the HIP kernel.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(_HERE))
SYNTH_LIB = os.path.join(_HERE, "libmot_synth.so")
EGO_FIXTURE = os.path.join(ROOT, "tests", "golden", "ego_drive0005.npz")


def build_lib(force: bool = False) -> str:
    """synth.hip -> libmot_synth.so (hipcc, gfx950; cross-compiles without a GPU). Built in-tree so that it travels to the GPU box."""
    import shutil
    import subprocess
    src = os.path.join(_HERE, "synth.hip")
    if not force and os.path.exists(SYNTH_LIB) and os.path.getmtime(SYNTH_LIB) >= os.path.getmtime(src):
        return SYNTH_LIB
    hipcc = next((c for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc") if c and os.path.exists(c)), None)
    if hipcc is None:
        raise RuntimeError("hipcc not found")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", src, "-o", SYNTH_LIB + ".tmp"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(SYNTH_LIB + ".tmp", SYNTH_LIB)
    return SYNTH_LIB
MAX_OBJECTS = 256

_KINDS = {  # half length, half width, height
    "car": (2.25, 0.9, 1.5), "ped": (0.3, 0.3, 1.7), "wall": (10.0, 0.15, 3.0), "pole": (0.15, 0.15, 3.0), "van": (2.6, 1.0, 2.1),
}


def load_ego(n_frames: int | None = None):
    """(v[F], yaw[F]): the ego motion of KITTI drive_0005 as the reference's second package ships it (154 frames);
    longer sequences continue the last speed and yaw rate."""
    d = np.load(EGO_FIXTURE)
    v, yaw = d["ego_v"].astype(np.float64), d["ego_yaw"].astype(np.float64)
    if n_frames is not None and n_frames > len(v):
        k = n_frames - len(v)
        rate = yaw[-1] - yaw[-2]
        v = np.concatenate([v, np.full(k, v[-1])]); yaw = np.concatenate([yaw, yaw[-1] + rate * np.arange(1, k + 1)])
    if n_frames is not None:
        v, yaw = v[:n_frames], yaw[:n_frames]
    return v, yaw


def ego_path(v, yaw, dt: float = 0.1):
    """world pose (x, y, heading) of the sensor per frame: the tracker's dead reckoning, integrated"""
    th = yaw - yaw[0]
    step = (dt * v)[:, None] * np.stack([np.cos(th), np.sin(th)], axis=1)
    step[0] = 0.0
    pos = np.cumsum(step, axis=0)
    return np.concatenate([pos, th[:, None], np.zeros((len(v), 1))], axis=1).astype(np.float32)


def street_scene(seed: int, path: np.ndarray, dt: float = 0.1, density: float = 1.0) -> np.ndarray:
    """objects [K, 8] (cx, cy, yaw, half length, half width, height, vx, vy) along the driven path; K <= MAX_OBJECTS.
    density scales the number of cars and pedestrians (the tracker-load knob)."""
    rng = np.random.default_rng(9_000_017 + seed)
    # arc length parametrisation of the path, extended by straight run-in / run-out
    p = path[:, :2].astype(np.float64); th = path[:, 2].astype(np.float64)
    pre = p[0] - np.outer(np.arange(40, 0, -1), [np.cos(th[0]), np.sin(th[0])])
    post = p[-1] + np.outer(np.arange(1, 61), [np.cos(th[-1]), np.sin(th[-1])])
    pts = np.concatenate([pre, p, post]); hd = np.concatenate([np.full(40, th[0]), th, np.full(60, th[-1])])
    seg = np.hypot(*np.diff(pts, axis=0).T); s = np.concatenate([[0.0], np.cumsum(seg)])
    keep = np.concatenate([[True], seg > 1e-6]); pts, hd, s = pts[keep], hd[keep], s[keep]
    L = s[-1]

    def at(sv, lat):
        x = np.interp(sv, s, pts[:, 0]); y = np.interp(sv, s, pts[:, 1]); h = np.interp(sv, s, hd)
        return x - lat * np.sin(h), y + lat * np.cos(h), h

    objs = []

    def add(kind, sv, lat, yaw_off=0.0, speed=0.0):
        hl, hw, h = _KINDS[kind]
        x, y, hh = at(sv, lat)
        yw = hh + yaw_off
        objs.append((x, y, yw, hl, hw, h, speed * np.cos(yw), speed * np.sin(yw)))

    for side in (-1, 1):                                     # parked cars / vans along both kerbs
        sv = rng.uniform(0, 6)
        while sv < L:
            if rng.random() < 0.75 * min(density, 1.3):
                add("van" if rng.random() < 0.15 else "car", sv, side * rng.uniform(3.6, 4.4), rng.normal(0, 0.03))
            sv += rng.uniform(5.5, 9.0) / max(density, 0.5)
    for _ in range(int(10 * density)):                       # traffic: same direction and oncoming
        oncoming = rng.random() < 0.5
        add("car", rng.uniform(0, L), (-1.8 if oncoming else 1.8) + rng.normal(0, 0.15), np.pi if oncoming else 0.0, rng.uniform(3.0, 10.0))
    for _ in range(int(22 * density)):                       # pedestrians on the pavements, a few crossing
        crossing = rng.random() < 0.15
        add("ped", rng.uniform(20, L - 20), rng.choice([-1, 1]) * rng.uniform(5.5, 8.0), np.pi / 2 * rng.choice([-1, 1]) if crossing else rng.choice([0.0, np.pi]),
            rng.uniform(0.4, 1.6))
    for side in (-1, 1):                                     # building fronts and street furniture
        sv = rng.uniform(0, 10)
        while sv < L:
            if rng.random() < 0.7:
                add("wall", sv, side * rng.uniform(10.0, 13.0), rng.normal(0, 0.02))
            sv += 22.0
        sv = rng.uniform(0, 15)
        while sv < L:
            add("pole", sv, side * rng.uniform(5.0, 5.4))
            sv += rng.uniform(18, 30)
    a = np.array(objs, np.float32)
    if len(a) > MAX_OBJECTS:   # keep the ones nearest to the driven part of the path
        mid = pts[len(pts) // 2]
        a = a[np.argsort(np.hypot(a[:, 0] - mid[0], a[:, 1] - mid[1]))[:MAX_OBJECTS]]
    out = np.zeros((MAX_OBJECTS, 8), np.float32)
    out[: len(a)] = a
    return out


def plaza_scene(seed: int, path: np.ndarray, dt: float = 0.1, rows=(5.5, 8.5, 11.5, 14.5, 17.5, 20.5), spacing: float = 6.8) -> np.ndarray:
    """objects [K, 8] for the TRACKER-LOAD scene (BASELINE.json configs[3] says "<= 64 tracks"; the street above never shows the tracker more than
    ~25 at a time, at any density: parked cars and house fronts shadow what stands behind them). An open square the ego vehicle drives through:
    nothing but standing and strolling people (0.6 x 0.6 x 1.7 m: every one passes the rule filter of box_fitting.cpp:97-158), in rows parallel to
    the driven path on both sides, `spacing` metres apart within a row, the rows staggered so that they rarely line up radially. A person shadows a
    few degrees of azimuth; 50-65 of the ~90 inside the 50 m region of interest return enough points for a box at any time."""
    rng = np.random.default_rng(5_000_011 + seed)
    p = path[:, :2].astype(np.float64); th = path[:, 2].astype(np.float64)
    pre = p[0] - np.outer(np.arange(40, 0, -1), [np.cos(th[0]), np.sin(th[0])])
    post = p[-1] + np.outer(np.arange(1, 61), [np.cos(th[-1]), np.sin(th[-1])])
    pts = np.concatenate([pre, p, post]); hd = np.concatenate([np.full(40, th[0]), th, np.full(60, th[-1])])
    seg = np.hypot(*np.diff(pts, axis=0).T); s = np.concatenate([[0.0], np.cumsum(seg)])
    keep = np.concatenate([[True], seg > 1e-6]); pts, hd, s = pts[keep], hd[keep], s[keep]
    L = s[-1]
    objs = []
    hl, hw, h = _KINDS["ped"]
    for side in (-1, 1):
        for r, lat in enumerate(rows):
            sv = rng.uniform(0, spacing) + 1.3 * r
            while sv < L:
                x = np.interp(sv, s, pts[:, 0]); y = np.interp(sv, s, pts[:, 1]); hh = np.interp(sv, s, hd)
                la = side * (lat + rng.uniform(-0.6, 0.6))
                strolling = rng.random() < 0.5
                yw = hh + (0.0 if rng.random() < 0.5 else np.pi)
                v = rng.uniform(0.3, 1.2) if strolling else 0.0
                objs.append((x - la * np.sin(hh), y + la * np.cos(hh), yw, hl, hw, h, v * np.cos(yw), v * np.sin(yw)))
                sv += spacing * rng.uniform(0.85, 1.15)
    a = np.array(objs, np.float32)
    if len(a) > MAX_OBJECTS:   # keep the ones nearest to the driven part of the path
        mid = pts[len(pts) // 2]
        a = a[np.argsort(np.hypot(a[:, 0] - mid[0], a[:, 1] - mid[1]))[:MAX_OBJECTS]]
    out = np.zeros((MAX_OBJECTS, 8), np.float32)
    out[: len(a)] = a
    return out


class SequenceRenderer:
    """renders [frames][scenes][stride] float4 clouds into HBM; torch tensors on `device`"""

    def __init__(self, device="cuda", lib_path: str | None = None):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        path = lib_path or SYNTH_LIB
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing — run `python __graft_entry__.py build` (or tools/synth/synth_dev.py build_lib())")
        self.lib = C.CDLL(path)
        self.lib.mot_synth_raycast.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                               C.c_ulonglong, C.c_void_p, C.c_void_p]

    def render(self, scene_ids, n_frames: int, n_points: int, stride: int, v=None, yaw=None, dt: float = 0.1, seed: int = 2025,
               density: float = 1.0, oversample: float = 1.25, frame_chunk: int = 32, scene: str = "street", order: str = "beam"):
        """-> (clouds [F][S][stride][4] float32 on the device, n [F][S] int32 numpy, objects list). Every frame is thinned
        uniformly (order preserved) to n_points returns; a frame with fewer returns keeps them all.
        order: the order of a frame's points in memory —
          "beam"    beam-major (a laser's whole revolution, then the next laser): KITTI's .bin files
          "firing"  azimuth-major (the 64 lasers of one firing, then the next azimuth step): what the velodyne driver's `velodyne_points`
                    carries, the topic the reference's ground node subscribes to (OT/src/groundremove/main.cpp:146)
          "random"  a seeded random permutation per frame (no locality at all: the worst case for anything that exploits the order)
        The SET of points of a frame is the same for every order (the thinning selects by ray id, not by position)."""
        if order not in ("beam", "firing", "random"):
            raise ValueError(f"order: {order!r}")
        torch = self.torch
        if v is None or yaw is None:
            v, yaw = load_ego(n_frames)
        path = ego_path(np.asarray(v[:n_frames], np.float64), np.asarray(yaw[:n_frames], np.float64), dt)
        S, F = len(scene_ids), n_frames
        n_az = -(-int(n_points * oversample) // 64)
        n_rays = 64 * n_az
        out = torch.empty((F, S, stride, 4), dtype=torch.float32, device=self.device)
        if stride > n_points:
            out[:, :, n_points:] = 0.0
        counts = torch.zeros((F, S), dtype=torch.int32, device=self.device)
        ego_d = torch.from_numpy(path).to(self.device).contiguous()
        stream = torch.cuda.current_stream().cuda_stream
        objs_all = []
        for si, sid in enumerate(scene_ids):
            objs = plaza_scene(int(sid), path, dt) if scene == "plaza" else street_scene(int(sid), path, dt, density)
            objs_all.append(objs)
            K = int((objs[:, 5] > 0).sum())
            objs_d = torch.from_numpy(objs[: max(K, 1)]).to(self.device).contiguous()
            for f0 in range(0, F, frame_chunk):
                fc = min(frame_chunk, F - f0)
                raw = torch.empty((fc, n_rays, 4), dtype=torch.float32, device=self.device)
                rc = self.lib.mot_synth_raycast(objs_d.data_ptr(), K, ego_d.data_ptr(), F, f0, fc, 1, int(sid), n_az, C.c_float(dt),
                                                C.c_ulonglong(seed), raw.data_ptr(), C.c_void_p(stream))
                if rc:
                    raise RuntimeError(f"mot_synth_raycast failed ({rc})")
                valid = ~torch.isnan(raw[:, :, 0])
                c = torch.cumsum(valid.to(torch.int64), dim=1)
                total = c[:, -1:].clamp(min=1)
                nt = torch.minimum(total, torch.full_like(total, n_points))
                j = (c * nt) // total
                jp = ((c - 1) * nt) // total
                sel = valid & (j > jp)           # which rays survive the thinning: decided in beam-major ray order, whatever the output order
                if order != "beam":
                    if order == "firing":        # ray = beam * n_az + a  ->  position a * 64 + beam
                        perm = torch.arange(n_rays, device=self.device, dtype=torch.int64).view(64, n_az).t().reshape(-1)
                        perm = perm[None].expand(fc, n_rays)
                    else:
                        gen = torch.Generator(device=self.device); gen.manual_seed(int(seed) * 1_000_003 + int(sid) * 1009 + f0)
                        perm = torch.randperm(n_rays, device=self.device, generator=gen)[None].expand(fc, n_rays)   # (one permutation per scene and chunk of frames)
                    raw = torch.gather(raw, 1, perm[:, :, None].expand(fc, n_rays, 4))
                    sel = torch.gather(sel, 1, perm)
                    j = torch.cumsum(sel.to(torch.int64), dim=1)   # output position (1-based) of every surviving ray in the new order
                fidx = torch.arange(f0, f0 + fc, device=self.device, dtype=torch.int64)[:, None]
                dest = ((fidx * S + si) * stride + (j - 1))[sel]
                out.view(-1, 4)[dest] = raw[sel]
                nf = torch.where(c[:, -1] > 0, nt[:, 0], torch.zeros_like(nt[:, 0])).to(torch.int32)
                counts[f0:f0 + fc, si] = nf
                if (nf < n_points).any():   # frames with fewer returns than asked for: zero their tails
                    for k in torch.nonzero(nf < n_points).flatten().tolist():
                        out[f0 + k, si, int(nf[k]):n_points] = 0.0
        torch.cuda.synchronize()
        return out, counts.cpu().numpy(), objs_all, path
