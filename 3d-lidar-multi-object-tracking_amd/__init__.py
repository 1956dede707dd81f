"""MI355X-native LiDAR perception hot path — Python host binding over the C-ABI (include/mot.h).

The product is ``libmot_hip.so`` (hand-written HIP for gfx950, built in-tree by ``build.py``); this module is
ctypes plumbing plus thin wrappers named after the reference functions they stand in for
(``groundRemove`` / ``componentClustering`` / ``boxFitting`` / ``getOriginPoints`` / ``immUkfJpdaf`` of
/root/reference/object_tracking). There is no CPU fallback: loading fails loudly when the extension is
missing, and ``Context()`` fails when no GPU is present.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmot_hip.so")

MOT_OK, MOT_E_ARG, MOT_E_CAPACITY, MOT_E_HIP, MOT_E_STATE = 0, 1, 2, 3, 4
MOT_MAX_BOXES_PER_FRAME = 1024   # include/mot.h
MOT_TRACKER_AUTO, MOT_TRACKER_SPLIT, MOT_TRACKER_STREAM = 0, 1, 2   # mot_set_tracker_mode (include/mot.h)
PRESET_OBJECT_TRACKING, PRESET_OBJECT_TRACKING0 = 0, 1
MASK_DROPPED, MASK_GROUND, MASK_ELEVATED = 0, 1, 2
NUM_CHANNEL, NUM_BIN = 80, 120


class MotParams(C.Structure):
    """struct mot_params (include/mot.h)"""
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
    """struct mot_track"""
    _fields_ = [("id", C.c_int32), ("track_manage", C.c_int32), ("is_static", C.c_int32), ("is_vis", C.c_int32),
                ("px", C.c_float), ("py", C.c_float), ("pz", C.c_float), ("lifetime", C.c_int32),
                ("v", C.c_double), ("yaw", C.c_double), ("vis_box", C.c_float * 24)]


class MotTrackState(C.Structure):
    """struct mot_track_state"""
    _fields_ = [("x_merge", C.c_double * 5), ("x_cv", C.c_double * 5), ("x_ctrv", C.c_double * 5), ("x_rm", C.c_double * 5),
                ("p_merge", C.c_double * 25), ("p_cv", C.c_double * 25), ("p_ctrv", C.c_double * 25), ("p_rm", C.c_double * 25),
                ("mode_prob", C.c_double * 3), ("z_pred", C.c_double * 6), ("s", C.c_double * 12), ("k", C.c_double * 30),
                ("init_meas", C.c_double * 2), ("dist_from_init", C.c_double), ("best_yaw", C.c_double),
                ("lifetime", C.c_int32), ("track_manage", C.c_int32), ("is_static", C.c_int32), ("is_vis", C.c_int32),
                ("has_best_box", C.c_int32), ("_pad", C.c_int32), ("bbox", C.c_float * 24), ("best_bbox", C.c_float * 24)]


class MotError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mot error {code}: {msg}")
        self.code = code


EXPORTS = (
    "mot_abi_version", "mot_params_preset", "mot_create", "mot_destroy", "mot_reset", "mot_last_error",
    "mot_synchronize", "mot_stream", "mot_ground_remove", "mot_cluster", "mot_box_fit", "mot_ego_update",
    "mot_track_step", "mot_track_get_state", "mot_frames_dev", "mot_sequence_dev", "mot_get_ground", "mot_get_clusters",
    "mot_get_boxes", "mot_get_tracks", "mot_export_tracks_dev", "mot_side_params_default", "mot_cluster_products", "mot_cluster_products_host", "mot_box_markers", "mot_decode_pointcloud2_dev", "mot_time_stage",
    "mot_ground_remove_pointcloud2", "mot_box_fit_resident", "mot_frame_pointcloud2",
    "mot_reset_slot", "mot_stream_snapshot_size", "mot_stream_save", "mot_stream_load", "mot_frames_host", "mot_frames_host_xyz", "mot_frames_host_pointcloud2", "mot_wait_uploads", "mot_host_alloc", "mot_host_free", "mot_fetch_tracks_async",
    "mot_track_steps_dev", "mot_profile_kernel", "mot_profile_read", "mot_get_params", "mot_set_fused_outputs", "mot_set_tracker_mode", "mot_set_trace_ranges", "mot_reset_tracks_slot", "mot_export_tracks_packed_dev", "mot_set_launch_graphs",
    "mot_cluster_node_frame", "mot_ground_node_frame",
    "mot_gather_unique_id", "mot_gather_create", "mot_gather_contribute", "mot_gather_result", "mot_gather_synchronize", "mot_gather_destroy", "mot_gather_last_error",
)
ABI_VERSION = 6
OUT_GROUND, OUT_MASK, OUT_LABELS = 1, 2, 4

_libs: dict[str, C.CDLL] = {}


def load_library(path: str | None = None) -> C.CDLL:
    """dlopen the HIP extension (no fallback: a missing library is an error)."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing — run `python __graft_entry__.py build` (hipcc, gfx950). "
                          "There is no CPU fallback for this library.")
    lib = C.CDLL(path)
    lib.mot_last_error.restype = C.c_char_p
    lib.mot_last_error.argtypes = [C.c_void_p]
    lib.mot_stream.restype = C.c_void_p
    lib.mot_destroy.restype = None
    lib.mot_gather_last_error.restype = C.c_char_p
    lib.mot_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    lib.mot_host_free.argtypes = [C.c_void_p]
    if lib.mot_abi_version() != ABI_VERSION:
        raise ImportError(f"{path}: ABI version {lib.mot_abi_version()}, this binding is for {ABI_VERSION} — rebuild the library")
    _libs[path] = lib
    return lib


def params(preset: int = PRESET_OBJECT_TRACKING, lib: C.CDLL | None = None, **overrides) -> MotParams:
    lib = lib or load_library()
    p = MotParams()
    rc = lib.mot_params_preset(preset, C.byref(p))
    if rc:
        raise MotError(rc, "mot_params_preset")
    for k, v in overrides.items():
        setattr(p, k, v)
    return p


class MotSideParams(C.Structure):
    """mirror of struct mot_side_params (include/mot.h)"""
    _fields_ = [("cell_size", C.c_float), ("cost_width", C.c_int32), ("cost_height", C.c_int32), ("cost_resolution", C.c_double),
                ("cost_offset_x", C.c_double), ("cost_offset_y", C.c_double), ("height_limit", C.c_double),
                ("car_length", C.c_double), ("car_width", C.c_double), ("cost_offset_z", C.c_double)]


class MotClusterFrame(C.Structure):
    """mirror of struct mot_cluster_frame (include/mot.h): counts + views into the context's page-locked block"""
    _fields_ = [("num_cluster", C.c_int32), ("n_clustered", C.c_int32), ("n_obstacles", C.c_int32), ("n_boxes", C.c_int32), ("n_undefined", C.c_int32),
                ("cost_cells", C.c_int32), ("clustered_xyzw", C.POINTER(C.c_float)), ("obstacles_xyzc", C.POINTER(C.c_float)), ("cost_map", C.POINTER(C.c_int32)),
                ("boxes", C.POINTER(C.c_float)), ("box_cluster", C.POINTER(C.c_int32)), ("centroid_extent", C.POINTER(C.c_float))]


def _pts(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 4:
        raise ValueError("points must be (n, 4) float32: x, y, z, w")
    return a


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """One mot_ctx: ``max_batch`` independent sensor-stream slots on one GPU, one HIP stream."""

    def __init__(self, p: MotParams | None = None, device: int = 0, max_points: int = 131072, max_batch: int = 1,
                 max_tracks_total: int = 4096, lib_path: str | None = None):
        self.lib = load_library(lib_path)
        self.params = p if p is not None else params(lib=self.lib)
        self.max_points, self.max_batch, self.max_tracks_total = max_points, max_batch, max_tracks_total
        h = C.c_void_p()
        rc = self.lib.mot_create(C.byref(self.params), device, max_points, max_batch, max_tracks_total, C.byref(h))
        if rc:
            raise MotError(rc, "mot_create failed (no GPU / bad arguments); this library has no CPU fallback")
        self._h = h

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if getattr(self, "_h", None):
            self.lib.mot_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc:
            raise MotError(rc, (self.lib.mot_last_error(self._h) or b"").decode())

    def synchronize(self):
        self._ck(self.lib.mot_synchronize(self._h))

    def reset(self):
        self._ck(self.lib.mot_reset(self._h))

    def reset_slot(self, slot: int):
        self._ck(self.lib.mot_reset_slot(self._h, slot))

    def reset_tracks_slot(self, slot: int):
        """forget the tracks of one stream, keep its ego pose (the origin of its global frame)"""
        self._ck(self.lib.mot_reset_tracks_slot(self._h, slot))

    def stream_save(self, slot: int = 0) -> bytes:
        """the tracker state of one stream as a relocatable block (mot_stream_save): load it into any slot of a context with the
        same max_tracks_total and the stream continues bit for bit"""
        cap = C.c_size_t(0)
        self._ck(self.lib.mot_stream_snapshot_size(self._h, C.byref(cap)))
        buf = (C.c_char * cap.value)(); n = C.c_size_t(0)
        self._ck(self.lib.mot_stream_save(self._h, slot, buf, cap, C.byref(n)))
        return bytes(buf[: n.value])

    def stream_load(self, slot: int, blob: bytes):
        self._ck(self.lib.mot_stream_load(self._h, slot, blob, C.c_size_t(len(blob))))

    def set_launch_graphs(self, on: bool = True):
        """send the fused entry points' launch sequence as one hipGraph launch (per-frame latency path; default off)"""
        self._ck(self.lib.mot_set_launch_graphs(self._h, int(on)))

    def set_trace_ranges(self, on: bool = True):
        """roctx ranges "mot:ground / cluster / box / tracker" around the stages of the fused entry points (rocprofv3 --marker-trace)"""
        self._ck(self.lib.mot_set_trace_ranges(self._h, int(on)))

    def set_tracker_mode(self, mode: int):
        """0 auto (by the number of streams per call), 1 split (four launches, tracks of all streams over the whole chip), 2 stream (one launch, a workgroup per stream)"""
        self._ck(self.lib.mot_set_tracker_mode(self._h, mode))

    def set_fused_outputs(self, flags: int):
        """which by-products of the ground stage the fused entry points write (OUT_GROUND | OUT_MASK | OUT_LABELS; default 0: on demand)"""
        self._ck(self.lib.mot_set_fused_outputs(self._h, flags))

    def _track_buffer(self, slot, max_tracks):
        """records a call can deliver: one per track EVER created on the stream, which outgrows the number of slots on a long run"""
        if max_tracks:
            return max_tracks
        hint = getattr(self, "_nt_hint", None)
        if hint is None:
            hint = self._nt_hint = {}
        return max(self.max_tracks_total, hint.get(slot, 0) + 256)

    def _ck_tracks(self, rc, n, cap):
        """mot_get_tracks / mot_track_step deliver the records AND report MOT_E_CAPACITY once a stream has used up max_tracks_total
        (sticky until reset): a soft condition here — the records are returned, `capacity_exceeded` says so"""
        if rc == MOT_E_CAPACITY and 0 <= n <= cap:   # (n = -1: mot_track_step refused the frame — more than 1024 boxes — and did not run: an error like any other)
            return True
        self._ck(rc)
        return False

    # ------------------------------------------------------------------ stage calls, host buffers
    def ground_remove(self, xyzw, want_mask: bool = True):
        """groundRemove(cloud, elevatedCloud, groundCloud) — OT/include/ground_removal.h:62-64"""
        a = _pts(xyzw); n = len(a)
        elev = np.empty((max(n, 1), 4), np.float32); ground = np.empty((max(n, 1), 4), np.float32)
        mask = np.zeros(max(n, 1), np.uint8) if want_mask else None
        ne, ng = C.c_int(0), C.c_int(0)
        self._ck(self.lib.mot_ground_remove(self._h, _vp(a), n, _vp(elev), C.byref(ne), _vp(ground), C.byref(ng), _vp(mask)))
        out = dict(elevated=elev[: ne.value].copy(), ground=ground[: ng.value].copy())
        if want_mask:
            out["mask"] = mask[:n].copy()
        return out

    def ground_remove_pointcloud2(self, payload, n: int, point_step: int, off_x: int, off_y: int, off_z: int, want_mask: bool = True):
        """fromROSMsg + groundRemove on a sensor_msgs/PointCloud2 payload in host memory (one H2D of the raw records)"""
        raw = np.ascontiguousarray(payload, np.uint8).reshape(-1)
        assert raw.size >= n * point_step
        elev = np.empty((max(n, 1), 4), np.float32); ground = np.empty((max(n, 1), 4), np.float32)
        mask = np.zeros(max(n, 1), np.uint8) if want_mask else None
        ne, ng = C.c_int(0), C.c_int(0)
        self._ck(self.lib.mot_ground_remove_pointcloud2(self._h, _vp(raw), n, point_step, off_x, off_y, off_z, _vp(elev), C.byref(ne),
                                                        _vp(ground), C.byref(ng), _vp(mask)))
        out = dict(elevated=elev[: ne.value].copy(), ground=ground[: ng.value].copy())
        if want_mask:
            out["mask"] = mask[:n].copy()
        return out

    def frame_pointcloud2(self, payload, n: int, point_step: int, off_x: int, off_y: int, off_z: int):
        """ground -> cluster -> box on a PointCloud2 payload in host memory, everything resident in slot 0 (asynchronous)"""
        raw = np.ascontiguousarray(payload, np.uint8).reshape(-1)
        assert raw.size >= n * point_step
        self._ck(self.lib.mot_frame_pointcloud2(self._h, _vp(raw), n, point_step, off_x, off_y, off_z))

    def box_fit_resident(self, max_boxes: int = 4096):
        """boxFitting on the cloud and label grid that cluster() left resident in slot 0"""
        boxes = np.zeros((max_boxes, 8, 3), np.float32); nb = C.c_int(0); bc = np.zeros(max_boxes, np.int32); nu = C.c_int(0)
        self._ck(self.lib.mot_box_fit_resident(self._h, _vp(boxes), max_boxes, C.byref(nb), _vp(bc), C.byref(nu)))
        return dict(boxes=boxes[: nb.value].copy(), box_cluster=bc[: nb.value].copy(), n_undefined=nu.value)

    def cluster(self, elevated_xyzw):
        """componentClustering(elevatedCloud, cartesianData, numCluster) — OT/include/component_clustering.h:20-22"""
        a = _pts(elevated_xyzw); n = len(a); G = self.params.num_grid
        grid = np.zeros((G, G), np.int32); nc = C.c_int(0); lab = np.zeros(max(n, 1), np.int32)
        self._ck(self.lib.mot_cluster(self._h, _vp(a), n, _vp(grid), C.byref(nc), _vp(lab)))
        return dict(grid=grid, num_cluster=nc.value, point_label=lab[:n].copy())

    def box_fit(self, elevated_xyzw, grid, num_cluster: int, max_boxes: int = 4096):
        """boxFitting(elevatedCloud, cartesianData, numCluster, ma) — OT/include/box_fitting.h:34-36"""
        a = _pts(elevated_xyzw); n = len(a)
        grid = np.ascontiguousarray(grid, np.int32)
        boxes = np.zeros((max_boxes, 8, 3), np.float32); nb = C.c_int(0); bc = np.zeros(max_boxes, np.int32); nu = C.c_int(0)
        self._ck(self.lib.mot_box_fit(self._h, _vp(a), n, _vp(grid), num_cluster, _vp(boxes), max_boxes, C.byref(nb), _vp(bc), C.byref(nu)))
        return dict(boxes=boxes[: nb.value].copy(), box_cluster=bc[: nb.value].copy(), n_undefined=nu.value)

    def ego_update(self, timestamp: float, v_gps: float, yaw_gps: float, slot: int = 0):
        """getOriginPoints(timestamp, originPoints, v_gps, yaw_gps) — OT/include/imm_ukf_jpda.h:15"""
        out = np.zeros(6)
        self._ck(self.lib.mot_ego_update(self._h, slot, C.c_double(timestamp), C.c_double(v_gps), C.c_double(yaw_gps), _vp(out)))
        return out

    def track_step(self, boxes_global, timestamp: float, slot: int = 0, max_tracks: int | None = None):
        """immUkfJpdaf(bBoxes, timestamp, ...) — OT/include/imm_ukf_jpda.h:19-22"""
        b = np.ascontiguousarray(boxes_global, np.float32).reshape(-1, 8, 3)
        cap = self._track_buffer(slot, max_tracks)
        arr = (MotTrack * cap)(); nt = C.c_int(0)
        rc = self.lib.mot_track_step(self._h, slot, _vp(b), len(b), C.c_double(timestamp), arr, cap, C.byref(nt))
        self._nt_hint[slot] = max(nt.value, 0)
        if rc == MOT_E_CAPACITY and nt.value > cap and not max_tracks:   # the step has run; only the buffer was too small: fetch again
            return self.get_tracks(slot)
        full = self._ck_tracks(rc, nt.value, cap)
        out = tracks_to_dict(arr, nt.value); out["capacity_exceeded"] = full
        return out

    def track_state(self, track_id: int, slot: int = 0):
        s = MotTrackState()
        self._ck(self.lib.mot_track_get_state(self._h, slot, track_id, C.byref(s)))
        return state_to_dict(s)

    # ------------------------------------------------------------------ fused frames, device buffers
    def frames_dev(self, d_ptr: int, frame_stride_floats: int, n_points, run_tracker: bool = False,
                   timestamps=None, ego_v=None, ego_yaw=None):
        """ground -> cluster -> box (-> tracker) for one frame per slot; everything stays in HBM. Asynchronous."""
        n = np.ascontiguousarray(n_points, np.int32); B = len(n)
        ts = np.ascontiguousarray(timestamps if timestamps is not None else np.zeros(B), np.float64)
        ev = np.ascontiguousarray(ego_v if ego_v is not None else np.zeros(B), np.float64)
        ey = np.ascontiguousarray(ego_yaw if ego_yaw is not None else np.zeros(B), np.float64)
        self._ck(self.lib.mot_frames_dev(self._h, C.c_void_p(d_ptr), C.c_long(frame_stride_floats), _vp(n), B,
                                         int(run_tracker), _vp(ts), _vp(ev), _vp(ey)))

    def sequence_dev(self, d_ptr: int, frame_stride_floats: int, n_points, timestamps, ego_v, ego_yaw, d_tracks_ptr: int = 0,
                     max_per_frame: int = 0, d_counts_ptr: int = 0):
        """SEQUENCE MODE (mot_sequence_dev): len(n_points) consecutive frames of ONE stream — stateless stages as one batch (slot k = frame
        k), the tracker chained on the device on stream 0's tracks. Optional device buffers receive the live tracks after every frame."""
        n = np.ascontiguousarray(n_points, np.int32); K = len(n)
        ts = np.ascontiguousarray(timestamps, np.float64); ev = np.ascontiguousarray(ego_v, np.float64); ey = np.ascontiguousarray(ego_yaw, np.float64)
        assert len(ts) == len(ev) == len(ey) == K
        self._ck(self.lib.mot_sequence_dev(self._h, C.c_void_p(d_ptr), C.c_long(frame_stride_floats), _vp(n), K, _vp(ts), _vp(ev), _vp(ey),
                                           C.c_void_p(d_tracks_ptr or None), int(max_per_frame), C.c_void_p(d_counts_ptr or None)))

    def get_ground(self, slot: int = 0, n_hint: int | None = None, want_clouds: bool = True):
        """n_hint: number of input points of the frame (length of the returned mask); the buffers are sized by max_points"""
        cap = self.max_points
        n = min(n_hint, cap) if n_hint is not None else cap
        elev = np.empty((cap, 4), np.float32) if want_clouds else None
        ground = np.empty((cap, 4), np.float32) if want_clouds else None
        mask = np.zeros(cap, np.uint8) if want_clouds else None
        ne, ng = C.c_int(0), C.c_int(0)
        self._ck(self.lib.mot_get_ground(self._h, slot, _vp(elev), C.byref(ne), _vp(ground), C.byref(ng), _vp(mask), cap))
        out = dict(n_elevated=ne.value, n_ground=ng.value)
        if want_clouds:
            out.update(elevated=elev[: ne.value].copy(), ground=ground[: ng.value].copy(), mask=mask[:n].copy())
        return out

    def get_clusters(self, slot: int = 0, n_elevated: int = 0):
        """n_elevated > 0: also return the per-point labels (the resident count is what is delivered; the hint only switches them on)"""
        G = self.params.num_grid
        grid = np.zeros((G, G), np.int32); nc = C.c_int(0)
        lab = np.zeros(self.max_points, np.int32) if n_elevated else None
        self._ck(self.lib.mot_get_clusters(self._h, slot, _vp(grid), C.byref(nc), _vp(lab), self.max_points if n_elevated else 0))
        return dict(grid=grid, num_cluster=nc.value, point_label=lab[:n_elevated].copy() if n_elevated else np.zeros(0, np.int32))

    def decode_pointcloud2_dev(self, d_data_ptr: int, n: int, point_step: int, off_x: int, off_y: int, off_z: int, off_w: int, d_xyzw_ptr: int):
        """PointCloud2 payload (device) -> float4 points (device), asynchronous on the context stream"""
        self._ck(self.lib.mot_decode_pointcloud2_dev(self._h, C.c_void_p(d_data_ptr), n, point_step, off_x, off_y, off_z, off_w, C.c_void_p(d_xyzw_ptr)))

    def cluster_products(self, slot: int = 0, sp: "MotSideParams | None" = None):
        """makeClusteredCloud / setObsMsg / createCostMap of the cluster node (component_clustering.cpp:311-379, 425-457) on the
        elevated cloud and label grid resident in ``slot``"""
        if sp is None:
            sp = MotSideParams(); self._ck(self.lib.mot_side_params_default(C.byref(sp)))
        n = self.max_points; G = self.params.num_grid
        cc = np.zeros((n, 4), np.float32); ob = np.zeros((G * G, 4), np.float32)
        cm = np.zeros(sp.cost_width * sp.cost_height, np.int32); ncc = C.c_int(0); nob = C.c_int(0)
        self._ck(self.lib.mot_cluster_products(self._h, slot, C.byref(sp), _vp(cc), n, C.byref(ncc), _vp(ob), G * G, C.byref(nob), _vp(cm)))
        return dict(clustered=cc[: ncc.value].copy(), obstacles=ob[: nob.value].copy(), cost_map=cm.reshape(sp.cost_height, sp.cost_width))

    def cluster_products_host(self, elev, grid, sp: "MotSideParams | None" = None):
        """the same on a caller-supplied cloud and label grid (the reference functions' argument lists)"""
        if sp is None:
            sp = MotSideParams(); self._ck(self.lib.mot_side_params_default(C.byref(sp)))
        a = _pts(elev); n = len(a); G = self.params.num_grid
        grid = np.ascontiguousarray(grid, np.int32)
        cc = np.zeros((max(n, 1), 4), np.float32); ob = np.zeros((G * G, 4), np.float32)
        cm = np.zeros(sp.cost_width * sp.cost_height, np.int32); ncc = C.c_int(0); nob = C.c_int(0)
        self._ck(self.lib.mot_cluster_products_host(self._h, _vp(a), n, _vp(grid), C.byref(sp), _vp(cc), max(n, 1), C.byref(ncc), _vp(ob), G * G,
                                                    C.byref(nob), _vp(cm)))
        return dict(clustered=cc[: ncc.value].copy(), obstacles=ob[: nob.value].copy(), cost_map=cm.reshape(sp.cost_height, sp.cost_width))

    def cluster_node_frame(self, elev, sp: "MotSideParams | None" = None, copy: bool = True):
        """the cluster node's whole callback in one call (mot_cluster_node_frame): labelling, side products, box fit, cubes"""
        if sp is None:
            sp = MotSideParams(); self._ck(self.lib.mot_side_params_default(C.byref(sp)))
        a = _pts(elev); fr = MotClusterFrame()
        self._ck(self.lib.mot_cluster_node_frame(self._h, _vp(a), len(a), C.byref(sp), C.byref(fr)))
        view = lambda ptr, shape, dt: (np.ctypeslib.as_array(ptr, shape=shape).view(dt) if shape[0] else np.zeros(shape, dt))
        out = dict(num_cluster=fr.num_cluster, n_undefined=fr.n_undefined,
                   clustered=view(fr.clustered_xyzw, (fr.n_clustered, 4), np.float32), obstacles=view(fr.obstacles_xyzc, (fr.n_obstacles, 4), np.float32),
                   cost_map=view(fr.cost_map, (fr.cost_cells,), np.int32).reshape(sp.cost_height, sp.cost_width),
                   boxes=view(fr.boxes, (fr.n_boxes, 24), np.float32).reshape(-1, 8, 3), box_cluster=view(fr.box_cluster, (fr.n_boxes,), np.int32),
                   cubes=view(fr.centroid_extent, (fr.n_boxes, 6), np.float32))
        return {k: (v.copy() if copy and isinstance(v, np.ndarray) else v) for k, v in out.items()}

    def ground_node_frame(self, cloud, copy: bool = True):
        """mot_ground_node_frame: groundRemove with the two clouds as views into the context's page-locked block"""
        a = _pts(cloud); pe, pg = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(); ne, ng = C.c_int(0), C.c_int(0)
        self._ck(self.lib.mot_ground_node_frame(self._h, _vp(a), len(a), C.byref(pe), C.byref(ne), C.byref(pg), C.byref(ng)))
        e = np.ctypeslib.as_array(pe, shape=(ne.value, 4)) if ne.value else np.zeros((0, 4), np.float32)
        g = np.ctypeslib.as_array(pg, shape=(ng.value, 4)) if ng.value else np.zeros((0, 4), np.float32)
        return dict(elevated=e.copy() if copy else e, ground=g.copy() if copy else g)

    def box_markers(self, slot: int = 0, max_boxes: int = 1024):
        """mark_cluster (box_fitting.cpp:161-209) of every box of ``slot``'s last box stage -> [n_boxes, 6]: centroid xyz, extent xyz"""
        out = np.zeros((max_boxes, 6), np.float32); nb = C.c_int(0)
        self._ck(self.lib.mot_box_markers(self._h, slot, _vp(out), max_boxes, C.byref(nb)))
        return out[: nb.value].copy()

    def get_boxes(self, slot: int = 0, max_boxes: int = 4096):
        boxes = np.zeros((max_boxes, 8, 3), np.float32); nb = C.c_int(0); bc = np.zeros(max_boxes, np.int32); nu = C.c_int(0)
        self._ck(self.lib.mot_get_boxes(self._h, slot, _vp(boxes), max_boxes, C.byref(nb), _vp(bc), C.byref(nu)))
        return dict(boxes=boxes[: nb.value].copy(), box_cluster=bc[: nb.value].copy(), n_undefined=nu.value)

    def get_tracks(self, slot: int = 0, max_tracks: int | None = None):
        cap = self._track_buffer(slot, max_tracks)
        arr = (MotTrack * cap)(); nt = C.c_int(0)
        rc = self.lib.mot_get_tracks(self._h, slot, arr, cap, C.byref(nt))
        self._nt_hint[slot] = max(nt.value, 0)
        if rc == MOT_E_CAPACITY and nt.value > cap and not max_tracks:
            cap = nt.value + 256
            arr = (MotTrack * cap)()
            rc = self.lib.mot_get_tracks(self._h, slot, arr, cap, C.byref(nt))
        full = self._ck_tracks(rc, nt.value, cap)
        out = tracks_to_dict(arr, nt.value); out["capacity_exceeded"] = full
        return out

    def frames_host(self, h_ptr: int, frame_stride_floats: int, n_points, run_tracker: bool = False,
                    timestamps=None, ego_v=None, ego_yaw=None):
        """frames_dev for frames in (page-locked) HOST memory: the upload of this batch overlaps the previous batch's kernels"""
        n = np.ascontiguousarray(n_points, np.int32); B = len(n)
        ts = np.ascontiguousarray(timestamps if timestamps is not None else np.zeros(B), np.float64)
        ev = np.ascontiguousarray(ego_v if ego_v is not None else np.zeros(B), np.float64)
        ey = np.ascontiguousarray(ego_yaw if ego_yaw is not None else np.zeros(B), np.float64)
        self._ck(self.lib.mot_frames_host(self._h, C.c_void_p(h_ptr), C.c_long(frame_stride_floats), _vp(n), B,
                                          int(run_tracker), _vp(ts), _vp(ev), _vp(ey)))

    def frames_host_xyz(self, h_ptr: int, frame_stride_floats: int, n_points, run_tracker: bool = False,
                        timestamps=None, ego_v=None, ego_yaw=None):
        """frames_host for packed {x, y, z} records (12 bytes a point; frame_stride_floats >= 3 n): pcl::PointXYZ has no 4th value, the host link carries 25 % less"""
        n = np.ascontiguousarray(n_points, np.int32); B = len(n)
        ts = np.ascontiguousarray(timestamps if timestamps is not None else np.zeros(B), np.float64)
        ev = np.ascontiguousarray(ego_v if ego_v is not None else np.zeros(B), np.float64)
        ey = np.ascontiguousarray(ego_yaw if ego_yaw is not None else np.zeros(B), np.float64)
        self._ck(self.lib.mot_frames_host_xyz(self._h, C.c_void_p(h_ptr), C.c_long(frame_stride_floats), _vp(n), B,
                                              int(run_tracker), _vp(ts), _vp(ev), _vp(ey)))

    def frames_host_pointcloud2(self, payloads, n_points, point_step: int, off_x: int, off_y: int, off_z: int, off_w: int = -1, run_tracker: bool = False,
                                timestamps=None, ego_v=None, ego_yaw=None):
        """frames_host for one sensor_msgs/PointCloud2 payload per stream (payloads: host addresses, or numpy uint8 arrays kept alive by the caller until wait_uploads)"""
        n = np.ascontiguousarray(n_points, np.int32); B = len(n)
        ptrs = (C.c_void_p * B)(*[(p.ctypes.data if hasattr(p, "ctypes") else int(p)) if p is not None else None for p in payloads])
        ts = np.ascontiguousarray(timestamps if timestamps is not None else np.zeros(B), np.float64)
        ev = np.ascontiguousarray(ego_v if ego_v is not None else np.zeros(B), np.float64)
        ey = np.ascontiguousarray(ego_yaw if ego_yaw is not None else np.zeros(B), np.float64)
        self._ck(self.lib.mot_frames_host_pointcloud2(self._h, ptrs, _vp(n), B, point_step, off_x, off_y, off_z, off_w, int(run_tracker), _vp(ts), _vp(ev), _vp(ey)))

    def wait_uploads(self):
        self._ck(self.lib.mot_wait_uploads(self._h))

    def fetch_tracks_async(self, batch: int, h_tracks_ptr: int, max_per_slot: int, h_counts_ptr: int):
        self._ck(self.lib.mot_fetch_tracks_async(self._h, batch, C.c_void_p(h_tracks_ptr), max_per_slot, C.c_void_p(h_counts_ptr)))

    def track_steps_dev(self, d_boxes_ptr: int, box_stride_floats: int, m, timestamps):
        """immUkfJpdaf for one frame of every slot, boxes (global frame) already on the device"""
        mm = np.ascontiguousarray(m, np.int32); ts = np.ascontiguousarray(timestamps, np.float64)
        self._ck(self.lib.mot_track_steps_dev(self._h, C.c_void_p(d_boxes_ptr), C.c_long(box_stride_floats), _vp(mm), len(mm), _vp(ts)))

    def profile_kernel(self, kernel_id: int, every: int = 1):
        """time every `every`-th launch of kernel `kernel_id` inside frames_dev / frames_host (0 = off); see profile_read"""
        self._ck(self.lib.mot_profile_kernel(self._h, kernel_id, every))

    def profile_read(self):
        mean, mn, mx, k = C.c_float(0), C.c_float(0), C.c_float(0), C.c_int(0)
        self._ck(self.lib.mot_profile_read(self._h, C.byref(mean), C.byref(mn), C.byref(mx), C.byref(k)))
        return dict(mean_ms=mean.value, min_ms=mn.value, max_ms=mx.value, samples=k.value)

    def export_tracks_dev(self, batch: int, d_tracks_ptr: int, max_per_slot: int, d_counts_ptr: int):
        """live tracks of every slot -> caller's device buffer (the block that is all-gathered across GPUs)"""
        self._ck(self.lib.mot_export_tracks_dev(self._h, batch, C.c_void_p(d_tracks_ptr), max_per_slot, C.c_void_p(d_counts_ptr)))

    def export_tracks_packed_dev(self, batch: int, d_block_ptr: int, block_bytes: int):
        """live tracks of every slot, packed (counts header + records back to back) -> caller's device block"""
        self._ck(self.lib.mot_export_tracks_packed_dev(self._h, batch, C.c_void_p(d_block_ptr), C.c_long(block_bytes)))

    def time_stage(self, stage: int, batch: int, iters: int) -> float:
        """average ms per iteration of one stage re-run on resident data, HIP events on the context stream"""
        ms = C.c_float(0)
        self._ck(self.lib.mot_time_stage(self._h, stage, batch, iters, C.byref(ms)))
        return ms.value


def tracks_to_dict(arr, n):
    buf = np.frombuffer(arr, dtype=np.dtype([("id", "i4"), ("track_manage", "i4"), ("is_static", "i4"), ("is_vis", "i4"),
                                             ("p", "f4", 3), ("lifetime", "i4"), ("v_yaw", "f8", 2), ("vis_box", "f4", 24)]),
                        count=n) if n else None
    if n == 0:
        return dict(n=0, track_manage=np.zeros(0, np.int32), is_static=np.zeros(0, np.int32), is_vis=np.zeros(0, np.int32),
                    lifetime=np.zeros(0, np.int32), p=np.zeros((0, 3), np.float32), v_yaw=np.zeros((0, 2)), vis_box=np.zeros((0, 24), np.float32))
    return dict(n=n, track_manage=buf["track_manage"].copy(), is_static=buf["is_static"].copy(), is_vis=buf["is_vis"].copy(),
                lifetime=buf["lifetime"].copy(), p=buf["p"].copy(), v_yaw=buf["v_yaw"].copy(), vis_box=buf["vis_box"].copy())


def state_to_dict(s: MotTrackState):
    d = {}
    for name, _ in MotTrackState._fields_:
        v = getattr(s, name)
        d[name] = np.array(v[:]) if hasattr(v, "__len__") else v
    return d


# ---------------------------------------------------------------------- reference-named convenience layer
_default_ctx: Context | None = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def groundRemove(cloud):
    """drop-in for groundRemove(); returns (elevatedCloud, groundCloud)"""
    r = default_context().ground_remove(cloud, want_mask=False)
    return r["elevated"], r["ground"]


def componentClustering(elevatedCloud):
    """drop-in for componentClustering(); returns (cartesianData, numCluster)"""
    r = default_context().cluster(elevatedCloud)
    return r["grid"], r["num_cluster"]


def boxFitting(elevatedCloud, cartesianData, numCluster):
    """drop-in for boxFitting(); returns the list of 8-corner boxes"""
    return default_context().box_fit(elevatedCloud, cartesianData, numCluster)["boxes"]


def getOriginPoints(timestamp, v_gps, yaw_gps):
    return default_context().ego_update(timestamp, v_gps, yaw_gps).reshape(2, 3)


def immUkfJpdaf(bBoxes, timestamp):
    return default_context().track_step(bBoxes, timestamp)


class NativeGather:
    """include/mot.h mot_gather_*: the per-tick all-gather of the live-track blocks of a rank's contexts, issued from C (RCCL over xGMI on GPUs; a device copy
    with one rank and no communicator). contribute(ci) is what context ci's issuing thread calls after its frame tick."""

    def __init__(self, ctxs, batch: int, capacity_records: int, world: int = 1, rank: int = 0, unique_id: bytes | None = None):
        self.ctxs, self.batch, self.cap, self.world, self.rank = list(ctxs), batch, capacity_records, world, rank
        self.lib = self.ctxs[0].lib
        arr = (C.c_void_p * len(self.ctxs))(*[cx._h for cx in self.ctxs])
        self._g = C.c_void_p()
        idb = (C.c_char * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        rc = self.lib.mot_gather_create(arr, len(self.ctxs), batch, capacity_records, world, rank, idb, C.byref(self._g))
        if rc != MOT_OK:
            raise MotError(rc, "mot_gather_create failed" + (" (no RCCL in this process)" if rc == MOT_E_STATE else ""))
        self.block = ((batch * 4 + 15) & ~15) + capacity_records * C.sizeof(MotTrack)

    @staticmethod
    def unique_id(lib) -> bytes:
        buf = (C.c_char * 128)()
        rc = lib.mot_gather_unique_id(buf)
        if rc != MOT_OK:
            raise MotError(rc, "mot_gather_unique_id failed (no RCCL in this process)")
        return bytes(buf.raw)

    def _ck(self, rc):
        if rc != MOT_OK:
            raise MotError(rc, (self.lib.mot_gather_last_error(self._g) or b"").decode())

    def contribute(self, ci: int):
        self._ck(self.lib.mot_gather_contribute(self._g, ci))

    def result(self):
        """(device pointer of [world][n_ctx] packed blocks, block bytes, tick, hipEvent_t handle) of the last completed tick"""
        d, nb, tick, ev = C.c_void_p(), C.c_long(), C.c_long(), C.c_void_p()
        self._ck(self.lib.mot_gather_result(self._g, C.byref(d), C.byref(nb), C.byref(tick), C.byref(ev)))
        return d.value, nb.value, tick.value, ev.value

    def synchronize(self):
        self._ck(self.lib.mot_gather_synchronize(self._g))

    def close(self):
        if self._g:
            self.lib.mot_gather_destroy(self._g); self._g = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
