"""Where a cluster-node callback's time goes (mot_cluster_node_frame on host buffers): wall clock per call, and — under
`rocprofv3 --kernel-trace --stats` — the kernels' own time per call.   python tools/time_node_frame.py [reps]"""
import ctypes as C
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m)
    return m


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    mot = _load("mot_amd", os.path.join(PKG, "__init__.py"))
    synth = _load("mot_amd.synth", os.path.join(os.path.dirname(PKG), "tools", "synth", "synth.py"))
    with mot.Context(device=0, max_points=131072, max_batch=1, max_tracks_total=64) as c:
        clouds = [synth.make_cloud(120000, s, 0) for s in (0, 1)]
        L, h = c.lib, c._h
        sp = mot.MotSideParams(); c._ck(L.mot_side_params_default(C.byref(sp)))
        fr = mot.MotClusterFrame(); pe, pg = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(); ne, ng = C.c_int(0), C.c_int(0)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        for a in clouds:
            elev = c.ground_remove(a, want_mask=False)["elevated"]
            tg, tc = [], []
            for r in range(reps):
                t0 = time.perf_counter()
                c._ck(L.mot_ground_node_frame(h, vp(a), len(a), C.byref(pe), C.byref(ne), C.byref(pg), C.byref(ng)))
                t1 = time.perf_counter()
                c._ck(L.mot_cluster_node_frame(h, vp(elev), len(elev), C.byref(sp), C.byref(fr)))
                t2 = time.perf_counter()
                tg.append(t1 - t0); tc.append(t2 - t1)
            print(f"{len(a)} points, {len(elev)} elevated, {fr.n_boxes} boxes, {fr.n_clustered} clustered, {fr.n_obstacles} obstacles: "
                  f"mot_ground_node_frame {np.median(tg) * 1e3:.3f} ms, mot_cluster_node_frame {np.median(tc) * 1e3:.3f} ms (median of {reps})")


if __name__ == "__main__":
    main()
