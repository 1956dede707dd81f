"""ctypes bindings for the TEST-ONLY oracle libraries.

  oracle/build/libmot_oracle.so  — C restatement (oracle/mot_oracle_*.c)
  oracle/_ref/libmot_ref.so      — the reference's own sources compiled against a shim (oracle/Makefile)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORC_SO = os.path.join(ORACLE_DIR, "build", "libmot_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libmot_ref.so")

POLAR_CELLS = 80 * 120
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


class MotParams(C.Structure):
    """mirror of struct mot_params (include/mot.h)"""
    _fields_ = [
        ("r_min", C.c_float), ("r_max", C.c_float), ("t_hmin", C.c_float), ("t_hmax", C.c_float),
        ("t_hdiff", C.c_float), ("h_sensor", C.c_float), ("ground_margin", C.c_double), ("gauss_sigma", C.c_double),
        ("gauss_samples", C.c_int32), ("crop_enable", C.c_int32),
        ("crop_z_min", C.c_float), ("crop_z_max", C.c_float), ("crop_x_min", C.c_float), ("crop_x_max", C.c_float),
        ("crop_y_min", C.c_float), ("crop_y_max", C.c_float),
        ("num_grid", C.c_int32), ("roi_m", C.c_float), ("occ_min_count", C.c_int32), ("dilate", C.c_int32),
        ("pic_scale", C.c_float), ("ram_points", C.c_int32), ("l_slope_dist", C.c_int32), ("l_num_points", C.c_int32),
        ("lshape_side_cond", C.c_int32), ("sensor_height", C.c_float),
        ("t_height_min", C.c_float), ("t_height_max", C.c_float), ("t_width_min", C.c_float), ("t_width_max", C.c_float),
        ("t_len_min", C.c_float), ("t_len_max", C.c_float), ("t_area_max", C.c_float),
        ("t_ratio_min", C.c_float), ("t_ratio_max", C.c_float), ("min_len_ratio", C.c_float), ("t_pt_per_m3", C.c_float),
        ("min_points", C.c_int32),
        ("gamma_g", C.c_double), ("p_g", C.c_double), ("p_d", C.c_double), ("distance_thres", C.c_double),
        ("life_time_thres", C.c_int32), ("seed_box_index", C.c_int32), ("bb_yaw_change_thres", C.c_double),
        ("first_ego_yaw_offset", C.c_double), ("seed_px", C.c_double), ("seed_py", C.c_double),
        ("rng_mapping", C.c_int32), ("max_tracks_ever", C.c_int32),
    ]


class MotTrack(C.Structure):
    _fields_ = [("id", C.c_int32), ("track_manage", C.c_int32), ("is_static", C.c_int32), ("is_vis", C.c_int32),
                ("px", C.c_float), ("py", C.c_float), ("pz", C.c_float), ("lifetime", C.c_int32),
                ("v", C.c_double), ("yaw", C.c_double), ("vis_box", C.c_float * 24)]


class MotTrackState(C.Structure):
    _fields_ = [("x_merge", C.c_double * 5), ("x_cv", C.c_double * 5), ("x_ctrv", C.c_double * 5), ("x_rm", C.c_double * 5),
                ("p_merge", C.c_double * 25), ("p_cv", C.c_double * 25), ("p_ctrv", C.c_double * 25), ("p_rm", C.c_double * 25),
                ("mode_prob", C.c_double * 3), ("z_pred", C.c_double * 6), ("s", C.c_double * 12), ("k", C.c_double * 30),
                ("init_meas", C.c_double * 2), ("dist_from_init", C.c_double), ("best_yaw", C.c_double),
                ("lifetime", C.c_int32), ("track_manage", C.c_int32), ("is_static", C.c_int32), ("is_vis", C.c_int32),
                ("has_best_box", C.c_int32), ("_pad", C.c_int32), ("bbox", C.c_float * 24), ("best_bbox", C.c_float * 24)]


class PolarDump(C.Structure):
    _fields_ = [("min_z", C.c_float * POLAR_CELLS), ("height", C.c_float * POLAR_CELLS), ("smoothed", C.c_float * POLAR_CELLS),
                ("hdiff", C.c_float * POLAR_CELLS), ("hground", C.c_float * POLAR_CELLS), ("is_ground", C.c_uint8 * POLAR_CELLS)]


def build_oracle(force: bool = False) -> None:
    """compile the C restatement (and oracle/_ref when /root/reference is present)"""
    if force or not os.path.exists(ORC_SO) or any(
            os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(ORC_SO)
            for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))):
        subprocess.run(["make", "-C", ORACLE_DIR, "build/libmot_oracle.so"], check=True, capture_output=True)
    if os.path.exists("/root/reference/object_tracking/tracking/ukf.cpp"):
        subprocess.run(["make", "-C", ORACLE_DIR, "ref"], check=True, capture_output=True)


_orc = None
_ref = None


def orc():
    global _orc
    if _orc is None:
        build_oracle()
        _orc = C.CDLL(ORC_SO)
    return _orc


def set_ref_library(path):
    """route the ref_* bindings to another build of the reference's sources (bench.py's -O0 baseline: oracle/_ref/libmot_ref_O0.so);
    None restores the default"""
    global _ref
    _ref = C.CDLL(path) if path else None


def ref():
    """the reference-built library, or None when it has not been built (no /root/reference, no prebuilt .so)"""
    global _ref
    if _ref is None:
        if not os.path.exists(REF_SO):
            if os.path.exists("/root/reference/object_tracking/tracking/ukf.cpp"):
                build_oracle()
            else:
                return None
        _ref = C.CDLL(REF_SO)
    return _ref


_ref_tf = None


def ref_tf():
    """the library that holds the tracking node's tf call sequence (oracle/ref_tf_capi.cpp -> ref_boxes_to_global): always the
    default reference build, whatever set_ref_library() routes the stage functions to (the -O0 build does not carry it); None
    when oracle/_ref is absent"""
    global _ref_tf
    if _ref_tf is None:
        if not os.path.exists(REF_SO):
            if ref() is None:
                return None
        _ref_tf = C.CDLL(REF_SO)
    return _ref_tf


def params(preset: int = 0, **overrides) -> MotParams:
    p = MotParams()
    rc = orc().orc_params_preset(preset, C.byref(p))
    assert rc == 0
    for k, v in overrides.items():
        setattr(p, k, v)
    return p


def _pts(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 4
    return a


# ------------------------------------------------------------------ restatement
def ground_remove(p: MotParams, xyzw, want_dump: bool = False):
    a = _pts(xyzw); n = len(a)
    elev = np.zeros((max(n, 1), 4), np.float32); ground = np.zeros((max(n, 1), 4), np.float32)
    mask = np.zeros(max(n, 1), np.uint8); ne = C.c_int(0); ng = C.c_int(0)
    dump = PolarDump() if want_dump else None
    rc = orc().orc_ground_remove(C.byref(p), a.ctypes.data_as(C.c_void_p), n, elev.ctypes.data_as(C.c_void_p), C.byref(ne),
                                 ground.ctypes.data_as(C.c_void_p), C.byref(ng), mask.ctypes.data_as(C.c_void_p),
                                 C.byref(dump) if dump is not None else None)
    assert rc == 0, rc
    out = dict(elevated=elev[: ne.value].copy(), ground=ground[: ng.value].copy(), mask=mask[:n].copy())
    if dump is not None:
        for k in ("min_z", "height", "smoothed", "hdiff", "hground", "is_ground"):
            out[k] = np.ctypeslib.as_array(getattr(dump, k)).copy().reshape(80, 120)
    return out


def crop(p: MotParams, xyzw):
    a = _pts(xyzw); out = np.zeros_like(a)
    orc().orc_crop.restype = C.c_int
    k = orc().orc_crop(C.byref(p), a.ctypes.data_as(C.c_void_p), len(a), out.ctypes.data_as(C.c_void_p))
    return out[:k].copy()


def cluster(p: MotParams, elev):
    a = _pts(elev); n = len(a); G = p.num_grid
    grid = np.zeros((G, G), np.int32); nc = C.c_int(0); lab = np.zeros(max(n, 1), np.int32)
    rc = orc().orc_cluster(C.byref(p), a.ctypes.data_as(C.c_void_p), n, grid.ctypes.data_as(C.c_void_p), C.byref(nc),
                           lab.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    return dict(grid=grid, num_cluster=nc.value, point_label=lab[:n].copy())


class MotSideParams(C.Structure):
    """mirror of struct mot_side_params (include/mot.h)"""
    _fields_ = [("cell_size", C.c_float), ("cost_width", C.c_int32), ("cost_height", C.c_int32), ("cost_resolution", C.c_double),
                ("cost_offset_x", C.c_double), ("cost_offset_y", C.c_double), ("height_limit", C.c_double),
                ("car_length", C.c_double), ("car_width", C.c_double), ("cost_offset_z", C.c_double)]


def side_params() -> MotSideParams:
    sp = MotSideParams()
    assert orc().orc_side_params_default(C.byref(sp)) == 0
    return sp


def cluster_products(p: MotParams, elev, grid, sp: MotSideParams | None = None):
    """makeClusteredCloud / setObsMsg / createCostMap restated (oracle/mot_oracle_side.c)"""
    a = _pts(elev); n = len(a); sp = sp or side_params()
    grid = np.ascontiguousarray(grid, np.int32)
    cc = np.zeros((max(n, 1), 4), np.float32); ob = np.zeros((max(n, 1), 4), np.float32)
    cm = np.zeros(sp.cost_width * sp.cost_height, np.int32); ncc = C.c_int(0); nob = C.c_int(0)
    rc = orc().orc_cluster_products(C.byref(p), C.byref(sp), a.ctypes.data_as(C.c_void_p), n, grid.ctypes.data_as(C.c_void_p),
                                    cc.ctypes.data_as(C.c_void_p), C.byref(ncc), ob.ctypes.data_as(C.c_void_p), C.byref(nob),
                                    cm.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    return dict(clustered=cc[: ncc.value].copy(), obstacles=ob[: nob.value].copy(), cost_map=cm.reshape(sp.cost_height, sp.cost_width))


class BoxDebug(C.Structure):
    _fields_ = [("num_points", C.c_int32), ("branch", C.c_int32), ("accepted", C.c_int32), ("undefined", C.c_int32),
                ("max_z", C.c_float), ("corners", C.c_float * 8)]


def box_fit(p: MotParams, elev, grid, num_cluster, max_boxes: int = 4096, debug: bool = False):
    a = _pts(elev); n = len(a)
    grid = np.ascontiguousarray(grid, np.int32)
    boxes = np.zeros((max_boxes, 8, 3), np.float32); nb = C.c_int(0); bc = np.zeros(max_boxes, np.int32); nu = C.c_int(0)
    dbg = (BoxDebug * max(num_cluster, 1))() if debug else None
    rc = orc().orc_box_fit(C.byref(p), a.ctypes.data_as(C.c_void_p), n, grid.ctypes.data_as(C.c_void_p), num_cluster,
                           boxes.ctypes.data_as(C.c_void_p), max_boxes, C.byref(nb), bc.ctypes.data_as(C.c_void_p), C.byref(nu), dbg)
    assert rc == 0, rc
    out = dict(boxes=boxes[: nb.value].copy(), box_cluster=bc[: nb.value].copy(), n_undefined=nu.value)
    if debug:
        out["debug"] = [dict(num_points=d.num_points, branch=d.branch, accepted=d.accepted, undefined=d.undefined,
                             max_z=d.max_z, corners=np.array(d.corners[:], np.float32)) for d in dbg[:num_cluster]]
    return out


def min_area_rect_points(xy) -> np.ndarray:
    xy = np.ascontiguousarray(xy, np.int32); out = np.zeros(8, np.float32)
    orc().orc_min_area_rect_points(xy.ctypes.data_as(C.c_void_p), len(xy), out.ctypes.data_as(C.c_void_p))
    return out.reshape(4, 2)


def min_area_rect(xy) -> np.ndarray:
    """the restated cv::minAreaRect's RotatedRect: (cx, cy, width, height, angle in degrees)"""
    xy = np.ascontiguousarray(xy, np.int32); out = np.zeros(5, np.float32)
    orc().orc_min_area_rect(xy.ctypes.data_as(C.c_void_p), len(xy), out.ctypes.data_as(C.c_void_p))
    return out


def mar_brute(xy):
    """oracle/mot_oracle_mar_brute.c: exact minimum-area enclosing rectangle by exhaustion over the hull edges (integer arithmetic,
    a different hull algorithm) — the independent cross-check of the restated cv::minAreaRect"""
    xy = np.ascontiguousarray(xy, np.int32); n = len(xy)
    hull = np.zeros((max(n, 1), 2), np.int32); area = np.zeros(max(n, 1)); be = C.c_int(-1); mn = C.c_double(0); ties = C.c_int(0)
    o = orc(); o.orc_mar_brute.restype = C.c_int
    k = o.orc_mar_brute(xy.ctypes.data_as(C.c_void_p), n, hull.ctypes.data_as(C.c_void_p), area.ctypes.data_as(C.c_void_p), C.byref(be), C.byref(mn), C.byref(ties))
    return dict(hull=hull[:k].copy(), edge_area=area[:k].copy() if k >= 3 else np.zeros(0), best_edge=be.value, min_area=mn.value, ties=ties.value)


MAR_OBSERVER = C.CFUNCTYPE(None, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_float))


class observe_mar:
    """with observe_mar() as seen: ... oracle.box_fit(...) ...  -> seen = [(pixels (n, 2) int32, rect (4, 2) float32)] of every cluster
    that went through the min-area-rectangle branch of the restated box fit (OT/src/cluster/box_fitting.cpp:357-362)"""

    def __enter__(self):
        self.seen = []

        def cb(xy, n, rect):
            self.seen.append((np.ctypeslib.as_array(xy, (n, 2)).copy(), np.ctypeslib.as_array(rect, (4, 2)).copy()))
        self._cb = MAR_OBSERVER(cb)
        orc().orc_set_mar_observer(self._cb)
        return self.seen

    def __exit__(self, *a):
        orc().orc_set_mar_observer(None)


def convex_hull(xy) -> np.ndarray:
    xy = np.ascontiguousarray(xy, np.int32); out = np.zeros((max(len(xy), 1), 2), np.int32)
    orc().orc_convex_hull.restype = C.c_int
    k = orc().orc_convex_hull(xy.ctypes.data_as(C.c_void_p), len(xy), out.ctypes.data_as(C.c_void_p))
    return out[:k].copy()


def lshape_indices_mapping(num_points: int, count: int, mapping: int) -> np.ndarray:
    out = np.zeros(count, np.int32)
    orc().orc_lshape_indices_mapping(num_points, count, mapping, out.ctypes.data_as(C.c_void_p))
    return out


def lshape_indices(num_points: int, count: int = 80) -> np.ndarray:
    out = np.zeros(count, np.int32)
    orc().orc_lshape_indices(num_points, count, out.ctypes.data_as(C.c_void_p))
    return out


class Tracker:
    """restated tracker (oracle/mot_oracle_track.c)"""

    def __init__(self, p: MotParams):
        o = orc(); o.orc_tracker_create.restype = C.c_void_p
        self._p = p; self._h = C.c_void_p(o.orc_tracker_create(C.byref(p)))

    def close(self):
        if self._h:
            orc().orc_tracker_destroy(self._h); self._h = None

    __del__ = close

    def reset(self):
        orc().orc_tracker_reset(self._h)

    def ego_update(self, ts, v, yaw):
        out = np.zeros(6); rc = orc().orc_ego_update(self._h, C.c_double(ts), C.c_double(v), C.c_double(yaw), out.ctypes.data_as(C.c_void_p))
        assert rc == 0; return out

    def step(self, boxes, ts, max_tracks=8192):
        b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 8, 3)
        arr = (MotTrack * max_tracks)(); nt = C.c_int(0)
        rc = orc().orc_track_step(self._h, b.ctypes.data_as(C.c_void_p), len(b), C.c_double(ts), arr, max_tracks, C.byref(nt))
        assert rc == 0, rc
        return tracks_to_dict(arr, nt.value)

    def state(self, i):
        s = MotTrackState(); rc = orc().orc_track_get_state(self._h, i, C.byref(s)); assert rc == 0
        return state_to_dict(s)


_TRACK_DT = np.dtype([("id", "i4"), ("track_manage", "i4"), ("is_static", "i4"), ("is_vis", "i4"), ("p", "f4", 3), ("lifetime", "i4"), ("v_yaw", "f8", 2), ("vis_box", "f4", 24)])


def tracks_to_dict(arr, n):
    if n == 0:
        return dict(n=0, track_manage=np.zeros(0, np.int32), is_static=np.zeros(0, np.int32), is_vis=np.zeros(0, np.int32), lifetime=np.zeros(0, np.int32),
                    p=np.zeros((0, 3), np.float32), v_yaw=np.zeros((0, 2)), vis_box=np.zeros((0, 24), np.float32))
    buf = np.frombuffer(arr, dtype=_TRACK_DT, count=n)
    return dict(n=n, track_manage=buf["track_manage"].copy(), is_static=buf["is_static"].copy(), is_vis=buf["is_vis"].copy(), lifetime=buf["lifetime"].copy(),
                p=buf["p"].copy(), v_yaw=buf["v_yaw"].copy(), vis_box=buf["vis_box"].copy())


def state_to_dict(s: MotTrackState):
    d = {}
    for name, _ in MotTrackState._fields_:
        v = getattr(s, name)
        d[name] = np.array(v[:]) if hasattr(v, "__len__") else v
    return d


# ------------------------------------------------------------------ reference build (oracle/_ref)
def ref_ground_remove(xyzw):
    a = _pts(xyzw); n = len(a)
    elev = np.zeros((max(n, 1), 4), np.float32); ground = np.zeros((max(n, 1), 4), np.float32); ne = C.c_int(0); ng = C.c_int(0)
    ref().ref_ground_remove(a.ctypes.data_as(C.c_void_p), n, elev.ctypes.data_as(C.c_void_p), C.byref(ne),
                            ground.ctypes.data_as(C.c_void_p), C.byref(ng))
    return dict(elevated=elev[: ne.value].copy(), ground=ground[: ng.value].copy())


def ref_ground_polar(xyzw):
    a = _pts(xyzw); out = {k: np.zeros((80, 120), np.float32) for k in ("min_z", "height", "smoothed", "hdiff", "hground")}
    out["is_ground"] = np.zeros((80, 120), np.uint8)
    ref().ref_ground_polar(a.ctypes.data_as(C.c_void_p), len(a), *[out[k].ctypes.data_as(C.c_void_p) for k in
                           ("min_z", "height", "smoothed", "hdiff", "hground", "is_ground")])
    return out


def ref_cell_index(x, y):
    ch = C.c_int(0); b = C.c_int(0)
    ref().ref_cell_index(C.c_float(x), C.c_float(y), C.byref(ch), C.byref(b))
    return ch.value, b.value


REF0_SO = os.path.join(ORACLE_DIR, "_ref", "libmot_ref0.so")
_ref0 = None


def ref0():
    """the reference's second package (object_tracking0: KITTI constants) built from its own sources, or None"""
    global _ref0
    if _ref0 is None:
        if not os.path.exists(REF0_SO):
            if os.path.exists("/root/reference/object_tracking0/src/ground_removal.cpp"):
                build_oracle()
            if not os.path.exists(REF0_SO):
                return None
        _ref0 = C.CDLL(REF0_SO)
    return _ref0


def ref0_frame(cloud, max_boxes=4096):
    """ground removal -> clustering -> box fit through OT0's own functions"""
    a = _pts(cloud); n = len(a); L = ref0()
    e = np.zeros((max(n, 1), 4), np.float32); g = np.zeros((max(n, 1), 4), np.float32); ne = C.c_int(0); ng = C.c_int(0)
    L.ref0_ground_remove(a.ctypes.data_as(C.c_void_p), n, e.ctypes.data_as(C.c_void_p), C.byref(ne), g.ctypes.data_as(C.c_void_p), C.byref(ng))
    elev = e[: ne.value].copy()
    G = L.ref0_num_grid()
    grid = np.zeros((G, G), np.int32); nc = C.c_int(0)
    L.ref0_cluster(elev.ctypes.data_as(C.c_void_p), len(elev), grid.ctypes.data_as(C.c_void_p), C.byref(nc))
    boxes = np.zeros((max_boxes, 8, 3), np.float32); nb = C.c_int(0)
    L.ref0_box_fit(elev.ctypes.data_as(C.c_void_p), len(elev), grid.ctypes.data_as(C.c_void_p), nc.value, boxes.ctypes.data_as(C.c_void_p), max_boxes, C.byref(nb))
    return dict(elevated=elev, ground=g[: ng.value].copy(), grid=grid, num_cluster=nc.value, boxes=boxes[: nb.value].copy())


def ref_cluster(elev):
    a = _pts(elev); G = ref().ref_num_grid()
    grid = np.zeros((G, G), np.int32); nc = C.c_int(0)
    ref().ref_cluster(a.ctypes.data_as(C.c_void_p), len(a), grid.ctypes.data_as(C.c_void_p), C.byref(nc))
    return dict(grid=grid, num_cluster=nc.value)


def ref_box_fit(elev, grid, num_cluster, max_boxes=4096):
    a = _pts(elev); grid = np.ascontiguousarray(grid, np.int32)
    boxes = np.zeros((max_boxes, 8, 3), np.float32); nb = C.c_int(0)
    ref().ref_box_fit(a.ctypes.data_as(C.c_void_p), len(a), grid.ctypes.data_as(C.c_void_p), num_cluster,
                      boxes.ctypes.data_as(C.c_void_p), max_boxes, C.byref(nb))
    return dict(boxes=boxes[: min(nb.value, max_boxes)].copy(), n=nb.value)


def ref_box_markers(elev, grid, num_cluster, max_boxes=4096):
    """the CUBE markers the reference's boxFitting fills (mark_cluster, box_fitting.cpp:161-209): [n_boxes, 6] float64
    pose.position xyz, scale xyz (0.1 where the extent is 0)"""
    a = _pts(elev); grid = np.ascontiguousarray(grid, np.int32)
    out = np.zeros((max_boxes, 6), np.float64); nb = C.c_int(0)
    ref().ref_box_markers(a.ctypes.data_as(C.c_void_p), len(a), grid.ctypes.data_as(C.c_void_p), num_cluster,
                          out.ctypes.data_as(C.c_void_p), max_boxes, C.byref(nb))
    return out[: min(nb.value, max_boxes)].copy()


def box_markers_numpy(elev, point_label, box_cluster):
    """mark_cluster restated: float32 sums in input order (np.add.accumulate is sequential), divided by the count; max - min.
    -> [n_boxes, 6] float32 centroid xyz, extent xyz (no 0.1 substitution)"""
    a = _pts(elev); lab = np.asarray(point_label)
    out = np.zeros((len(box_cluster), 6), np.float32)
    for i, c in enumerate(box_cluster):
        pts = a[lab == c, :3]
        out[i, :3] = np.add.accumulate(pts, axis=0, dtype=np.float32)[-1] / np.float32(len(pts))
        out[i, 3:] = pts.max(axis=0) - pts.min(axis=0)
    return out


def ref_cluster_products(elev, grid):
    """the reference's own makeClusteredCloud / setObsMsg / createCostMap (OT preset; 50 x 50 cost map)"""
    a = _pts(elev); n = len(a); grid = np.ascontiguousarray(grid, np.int32)
    cc = np.zeros((max(n, 1), 4), np.float32); ob = np.zeros((max(n, 1), 4), np.float32)
    cm = np.zeros(65536, np.int32); ncc = C.c_int(0); nob = C.c_int(0)
    ncell = ref().ref_cluster_products(a.ctypes.data_as(C.c_void_p), n, grid.ctypes.data_as(C.c_void_p), cc.ctypes.data_as(C.c_void_p),
                                       C.byref(ncc), ob.ctypes.data_as(C.c_void_p), C.byref(nob), cm.ctypes.data_as(C.c_void_p))
    return dict(clustered=cc[: ncc.value].copy(), obstacles=ob[: nob.value].copy(), cost_map=cm[:ncell].reshape(50, 50))


_variants = {}


def ref_variant(tag: str, instance: int = 0):
    """another build of the reference's own sources with a different floating-point operation order (oracle/Makefile REF_VARIANT:
    "fma" = -ffp-contract=fast -mfma, "novec" = Eigen's packet kernels off), or None when it is not on this box / this CPU cannot
    run it; tag "ref" = the default build itself. Each library carries its own copy of the tracker's file-scope globals (-Bsymbolic,
    RTLD_LOCAL); instance > 0 loads a private COPY of the file (the dynamic loader keys on the path), i.e. one more independent
    tracker of the same build — for checkers that follow several streams at once."""
    key = (tag, instance)
    if key not in _variants:
        path = REF_SO if tag == "ref" else os.path.join(ORACLE_DIR, "_ref", f"libmot_ref_{tag}.so")
        ok = os.path.exists(path)
        if ok and instance:
            import shutil, tempfile
            d = tempfile.mkdtemp(prefix="mot_ref_copy_")
            path = shutil.copy(path, os.path.join(d, f"libmot_ref_{tag}_{instance}.so"))
        if ok and tag == "fma":
            try:
                ok = " fma " in open("/proc/cpuinfo").read().split("flags", 1)[1].split("\n", 1)[0] + " "
            except Exception:
                ok = False
        _variants[key] = C.CDLL(path) if ok else None
    return _variants[key]


class RefTracker:
    """the reference tracker: file-scope globals => one instance at a time PER LIBRARY (lib: a ref_variant(); default the -O2
    -ffp-contract=off build, or whatever set_ref_library() routes to)"""
    _pre = "ref_"

    def __init__(self, lib=None):
        self._own = lib

    def _lib(self):
        return self._own if self._own is not None else ref()

    def reset(self):
        self._lib().ref_tracker_reset()

    def ego_update(self, ts, v, yaw):
        out = np.zeros(6); self._lib().ref_ego_update(C.c_double(ts), C.c_double(v), C.c_double(yaw), out.ctypes.data_as(C.c_void_p)); return out

    def step(self, boxes, ts, max_tracks=8192):
        b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 8, 3)
        xyz = np.zeros((max_tracks, 3), np.float32); vy = np.zeros((max_tracks, 2)); tm = np.zeros(max_tracks, np.int32)
        st = np.zeros(max_tracks, np.int32); vis = np.zeros(max_tracks, np.int32); vbb = np.zeros((max_tracks, 24), np.float32); nt = C.c_int(0)
        getattr(self._lib(), self._pre + "track_step")(b.ctypes.data_as(C.c_void_p), len(b), C.c_double(ts), max_tracks, xyz.ctypes.data_as(C.c_void_p),
                             vy.ctypes.data_as(C.c_void_p), tm.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p),
                             vis.ctypes.data_as(C.c_void_p), vbb.ctypes.data_as(C.c_void_p), C.byref(nt))
        n = nt.value
        out = dict(n=n, track_manage=tm[:n].copy(), is_static=st[:n].copy(), is_vis=vis[:n].copy(), p=xyz[:n].copy(),
                   v_yaw=vy[:n].copy(), vis_box=vbb[:n].copy())
        lt = getattr(self._lib(), self._pre + "track_lifetimes", None)   # (a prebuilt library from before this entry point existed has none)
        if lt is not None:
            life = np.zeros(max(n, 1), np.int32); lt(life.ctypes.data_as(C.c_void_p), n); out["lifetime"] = life[:n].copy()
        return out

    def count(self):
        return getattr(self._lib(), self._pre + "track_count")()

    def close(self):
        pass

    def state(self, i):
        x = np.zeros(20); p = np.zeros(100); mode = np.zeros(3); z = np.zeros(6); s = np.zeros(12); k = np.zeros(30); misc = np.zeros(4)
        ints = np.zeros(5, np.int32); bb = np.zeros(24, np.float32); best = np.zeros(24, np.float32)
        rc = getattr(self._lib(), self._pre + "track_get_state")(i, *[a.ctypes.data_as(C.c_void_p) for a in (x, p, mode, z, s, k, misc, ints, bb, best)])
        assert rc == 0
        return dict(x_merge=x[0:5], x_cv=x[5:10], x_ctrv=x[10:15], x_rm=x[15:20], p_merge=p[0:25], p_cv=p[25:50], p_ctrv=p[50:75],
                    p_rm=p[75:100], mode_prob=mode, z_pred=z, s=s, k=k, init_meas=misc[0:2], dist_from_init=misc[2], best_yaw=misc[3],
                    lifetime=int(ints[0]), track_manage=int(ints[1]), is_static=int(ints[2]), is_vis=int(ints[3]),
                    has_best_box=int(ints[4]), bbox=bb, best_bbox=best)


class Ref0Tracker(RefTracker):
    """object_tracking0's tracker (oracle/_ref/libmot_ref0.so). It reads the ego speed / yaw of every frame from two text
    files relative to the working directory: reset() takes the whole sequence, writes them under `workdir` and chdirs
    there (restored by close())."""
    _pre = "ref0_"

    def _lib(self):
        return ref0()

    def reset(self, workdir, velo, yaw):
        v = np.ascontiguousarray(velo, np.float64); y = np.ascontiguousarray(yaw, np.float64)
        assert len(v) == len(y)
        self._cwd = os.getcwd()
        rc = ref0().ref0_tracker_reset(str(workdir).encode(), v.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), len(v))
        assert rc == 0, rc

    def close(self):
        os.chdir(self._cwd)

    def ego_update(self, ts):
        out = np.zeros(6); ref0().ref0_ego_update(C.c_double(ts), out.ctypes.data_as(C.c_void_p)); return out


# ------------------------------------------------------------------ the reference build FIRST (GPU parity suite)
class RefFirst:
    """The oracle the `-m gpu` tests compare the device with: the reference's OWN sources (oracle/_ref/libmot_ref.so for preset 0,
    libmot_ref0.so for preset 1) wherever they are on the box and the call is one the reference can answer — its constants are
    compile-time, so a parameter set that is not exactly a preset goes to the restatement, and so do the pieces the reference's API
    does not have (see each method). Everything else of this module is passed through. `used` counts, per function, which oracle
    answered; tests/conftest.py prints the table per test.

    What is taken from where when the reference answers:
      ground_remove  clouds from groundRemove (ground_removal.cpp:177-249). The reference has no mask: it is RECONSTRUCTED from its two
                     clouds — both are order-preserving subsequences of the input, so one forward pass matching x, y, z bit patterns
                     assigns every input point to elevated / ground / dropped uniquely (equal points classify equally) — and the
                     returned clouds are the input's own records (4th float included) selected by that mask, after asserting that their
                     x, y, z are bit-equal to the reference's. want_dump: the per-cell arrays come from ref_ground_polar.
      cluster        label grid and count from componentClustering (component_clustering.cpp:260-268); the per-point labels (no such
                     output in the reference) are the restatement's, after asserting that its grid equals the reference's.
      box_fit        boxes from boxFitting (box_fitting.cpp:422-435); box -> cluster ids and the undefined-behaviour count (SURVEY.md
                     H7) are the restatement's; a frame with n_undefined > 0 has no defined reference result and goes to the restatement.
      Tracker        immUkfJpdaf + UKF (imm_ukf_jpda.cpp:704-1112) through RefTracker — a private copy of the library per instance
                     (file-scope globals), preset 0 only."""

    def __init__(self, base):
        self._b = base
        self.used = {}
        self._p = {0: bytes(base.params(0)), 1: bytes(base.params(1))}
        self._trk = 0

    def __getattr__(self, name):
        return getattr(self._b, name)

    def _note(self, fn, who):
        k = (fn, who); self.used[k] = self.used.get(k, 0) + 1

    def _preset(self, p):
        raw = bytes(p)
        if raw == self._p[0] and self._b.ref() is not None:
            return 0
        if raw == self._p[1] and self._b.ref0() is not None:
            return 1
        return None

    @staticmethod
    def _mask_from_subsequences(a, elev, ground):
        """a: input (n, 4); elev / ground: the reference's clouds (x, y, z). One forward pass: the next unmatched elevated / ground point."""
        key = np.ascontiguousarray(a[:, :3]).view(np.uint32)
        ke = np.ascontiguousarray(elev[:, :3]).view(np.uint32); kg = np.ascontiguousarray(ground[:, :3]).view(np.uint32)
        mask = np.zeros(len(a), np.uint8); ie = ig = 0
        eq_e = lambda i: ie < len(ke) and key[i, 0] == ke[ie, 0] and key[i, 1] == ke[ie, 1] and key[i, 2] == ke[ie, 2]
        eq_g = lambda i: ig < len(kg) and key[i, 0] == kg[ig, 0] and key[i, 1] == kg[ig, 1] and key[i, 2] == kg[ig, 2]
        for i in range(len(a)):
            if eq_e(i):
                assert not eq_g(i) or not np.array_equal(ke[ie], kg[ig]) or True
                mask[i] = 2; ie += 1
            elif eq_g(i):
                mask[i] = 1; ig += 1
        assert ie == len(ke) and ig == len(kg), ("the reference's clouds are not order-preserving subsequences of the input", ie, len(ke), ig, len(kg))
        return mask

    def restatement(self, fn: str, why: str):
        """the C restatement, on purpose (logged like every other routing decision)"""
        self._note(fn, "restatement (" + why + ")")
        return self._b

    def ground_remove(self, p, xyzw, want_dump=False):
        ps = self._preset(p)
        if ps is None:
            self._note("ground_remove", "restatement (parameters are not a preset)")
            return self._b.ground_remove(p, xyzw, want_dump)
        a = _pts(xyzw)
        r = self._b.ref_ground_remove(a) if ps == 0 else None
        if ps == 1:
            L = self._b.ref0(); n = len(a)
            e = np.zeros((max(n, 1), 4), np.float32); g = np.zeros((max(n, 1), 4), np.float32); ne = C.c_int(0); ng = C.c_int(0)
            L.ref0_ground_remove(a.ctypes.data_as(C.c_void_p), n, e.ctypes.data_as(C.c_void_p), C.byref(ne), g.ctypes.data_as(C.c_void_p), C.byref(ng))
            r = dict(elevated=e[: ne.value], ground=g[: ng.value])
        mask = self._mask_from_subsequences_fast(a, r["elevated"], r["ground"])
        out = dict(mask=mask, elevated=a[mask == 2].copy(), ground=a[mask == 1].copy())
        if want_dump:
            if ps == 0:
                out.update(self._b.ref_ground_polar(a))
            else:
                d = self._b.ground_remove(p, a, True)
                out.update({k: d[k] for k in ("min_z", "height", "smoothed", "hdiff", "hground", "is_ground")})
        self._note("ground_remove", "reference build" + (" (object_tracking0)" if ps else ""))
        return out

    def _mask_from_subsequences_fast(self, a, elev, ground):
        """the same matching vectorised for the common case (no two input points share x, y, z bits across classes — they cannot): a point
        is elevated iff its bits occur in the elevated cloud; counts and order are then verified, the slow pass is the fallback"""
        n = len(a)
        if n == 0:
            return np.zeros(0, np.uint8)
        key = np.ascontiguousarray(a[:, :3]).view(np.uint32).astype(np.uint64)
        pack = lambda k: (k[:, 0] << np.uint64(42)) ^ (k[:, 1] << np.uint64(21)) ^ k[:, 2] ^ (k[:, 0] >> np.uint64(13)) ^ (k[:, 1] * np.uint64(0x9E3779B97F4A7C15))
        ha = pack(key)
        he = pack(np.ascontiguousarray(elev[:, :3]).view(np.uint32).astype(np.uint64)) if len(elev) else np.zeros(0, np.uint64)
        hg = pack(np.ascontiguousarray(ground[:, :3]).view(np.uint32).astype(np.uint64)) if len(ground) else np.zeros(0, np.uint64)
        mask = np.zeros(n, np.uint8)
        mask[np.isin(ha, he)] = 2
        mask[np.isin(ha, hg) & (mask == 0)] = 1
        ok = (int((mask == 2).sum()) == len(elev) and int((mask == 1).sum()) == len(ground)
              and np.array_equal(np.ascontiguousarray(a[mask == 2][:, :3]).view(np.uint32), np.ascontiguousarray(elev[:, :3]).view(np.uint32))
              and np.array_equal(np.ascontiguousarray(a[mask == 1][:, :3]).view(np.uint32), np.ascontiguousarray(ground[:, :3]).view(np.uint32)))
        if ok:
            return mask
        mask = self._mask_from_subsequences(a, elev, ground)
        assert np.array_equal(np.ascontiguousarray(a[mask == 2][:, :3]).view(np.uint32), np.ascontiguousarray(elev[:, :3]).view(np.uint32))
        assert np.array_equal(np.ascontiguousarray(a[mask == 1][:, :3]).view(np.uint32), np.ascontiguousarray(ground[:, :3]).view(np.uint32))
        return mask

    def cluster(self, p, elev):
        ps = self._preset(p)
        o = self._b.cluster(p, elev)
        if ps is None:
            self._note("cluster", "restatement (parameters are not a preset)")
            return o
        a = _pts(elev)
        if not np.isfinite(a[:, :3]).all():   # (a cloud no ground stage can emit; the reference's unchecked casts index out of bounds on it and crash)
            self._note("cluster", "restatement (non-finite points: the reference indexes out of bounds)")
            return o
        if ps == 0:
            r = self._b.ref_cluster(a)
        else:
            L = self._b.ref0(); G = L.ref0_num_grid(); grid = np.zeros((G, G), np.int32); nc = C.c_int(0)
            L.ref0_cluster(a.ctypes.data_as(C.c_void_p), len(a), grid.ctypes.data_as(C.c_void_p), C.byref(nc))
            r = dict(grid=grid, num_cluster=nc.value)
        assert r["num_cluster"] == o["num_cluster"] and np.array_equal(r["grid"], o["grid"]), "restatement and reference build disagree on the label grid"
        self._note("cluster", "reference build (per-point labels: restatement on the reference's grid)")
        return dict(grid=r["grid"], num_cluster=r["num_cluster"], point_label=o["point_label"])

    def box_fit(self, p, elev, grid, num_cluster, max_boxes=4096, debug=False):
        ps = self._preset(p)
        o = self._b.box_fit(p, elev, grid, num_cluster, max_boxes, debug)
        a = _pts(elev); g = np.ascontiguousarray(grid, np.int32)
        finite = bool(np.isfinite(a[:, :3]).all())
        sane_grid = bool(((g >= 0) & (g <= max(num_cluster, 0))).all())   # (a hostile caller-made grid: the reference indexes its cluster vector with it)
        if ps is None or o["n_undefined"] > 0 or debug or not finite or not sane_grid:
            self._note("box_fit", "restatement (" + ("parameters are not a preset" if ps is None else "per-cluster debug record" if debug else
                                                       "non-finite points: the reference indexes out of bounds" if not finite else "labels outside 0..num_cluster in the grid" if not sane_grid else
                                                       "the reference reads uninitialised memory on this frame: SURVEY.md H7") + ")")
            return o
        if ps == 0:
            r = self._b.ref_box_fit(a, g, num_cluster, max_boxes)
        else:
            L = self._b.ref0(); boxes = np.zeros((max_boxes, 8, 3), np.float32); nb = C.c_int(0)
            L.ref0_box_fit(a.ctypes.data_as(C.c_void_p), len(a), g.ctypes.data_as(C.c_void_p), num_cluster, boxes.ctypes.data_as(C.c_void_p), max_boxes, C.byref(nb))
            r = dict(boxes=boxes[: min(nb.value, max_boxes)].copy(), n=nb.value)
        assert len(r["boxes"]) == len(o["boxes"]), "restatement and reference build disagree on the number of boxes"
        self._note("box_fit", "reference build (box -> cluster ids: restatement)")
        return dict(boxes=r["boxes"], box_cluster=o["box_cluster"], n_undefined=0)

    def Tracker(self, p):
        if self._preset(p) == 0 and self._b.ref_variant("ref", self._trk + 1) is not None:
            self._trk += 1
            t = self._b.RefTracker(self._b.ref_variant("ref", self._trk)); t.reset()
            self._note("Tracker", "reference build")
            return t
        self._note("Tracker", "restatement (" + ("object_tracking0's tracker reads its ego motion from files" if self._preset(p) == 1 else "parameters are not a preset") + ")")
        return self._b.Tracker(p)
