#!/usr/bin/env python3
"""Exploration runs on the REAL kernels (the CPU suite does the same on the emulator): randomised tracker sequences and large
irregular clouds through libmot_hip.so against the restatement, reporting — not asserting — what differs.
    python tests/explore_gpu.py [n_tracker_sequences] [n_clouds]
(lives in tests/ because it calls the oracle: test infrastructure, like everything that does)
Tracker: discrete outputs must match; continuous states are compared while the filter is well conditioned (see
tests/test_emu_tracker_random.py). A discrete mismatch on the GPU that the emulator does not show points at the device math
library (sin / cos / exp / atan2 differ from glibc in the last bit) meeting a threshold — worth a look, not necessarily a bug."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import load_pkg  # noqa: E402
import oracle_lib as O  # noqa: E402
import test_emu_large_random as LR  # noqa: E402
import test_emu_tracker_random as TR  # noqa: E402

mot = load_pkg()
n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_cloud = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for preset in (0, 1):
    p = O.params(preset)
    with mot.Context(mot.params(preset), max_points=4096, max_tracks_total=512) as c:
        worst = 0.0
        for seed in range(n_seq):
            c.reset(); T = O.Tracker(p)
            for f, (boxes, ts, v, yaw) in enumerate(TR.sequence(20000 + 1000 * preset + seed)):
                c.ego_update(ts, v, yaw); T.ego_update(ts, v, yaw)
                a = c.track_step(boxes, ts); o = T.step(boxes, ts)
                same = a["n"] == o["n"] and all(np.array_equal(a[k], o[k]) for k in ("track_manage", "is_static", "is_vis", "lifetime"))
                if not same:
                    bad += 1; print(f"tracker preset {preset} seed {seed} frame {f}: discrete outputs differ", a["track_manage"], o["track_manage"]); break
                for i in np.nonzero(o["track_manage"] > 0)[0]:
                    so = T.state(int(i))
                    if TR.well_conditioned(so):
                        sa = c.track_state(int(i))
                        worst = max(worst, float(np.abs(np.asarray(sa["x_merge"]) - so["x_merge"]).max() / max(np.abs(so["x_merge"]).max(), 1e-300)))
            T.close()
        print(f"tracker preset {preset}: {n_seq} sequences, worst relative state difference on well-conditioned tracks {worst:.3e}")
    with mot.Context(mot.params(preset), max_points=131072) as c:
        for seed in range(n_cloud):
            cloud = LR.big_cloud(30000 + 100 * preset + seed)
            g = c.ground_remove(cloud); og = O.ground_remove(p, cloud)
            ok = np.array_equal(g["mask"], og["mask"]) and np.array_equal(g["elevated"], og["elevated"])
            cl = c.cluster(og["elevated"]); ocl = O.cluster(p, og["elevated"])
            ok = ok and cl["num_cluster"] == ocl["num_cluster"] and np.array_equal(cl["grid"], ocl["grid"])
            if ok and ocl["num_cluster"] <= 4096:
                bx = c.box_fit_resident(); obx = O.box_fit(p, og["elevated"], ocl["grid"], ocl["num_cluster"])
                ok = np.array_equal(bx["boxes"].view(np.uint32), obx["boxes"].view(np.uint32))
            if not ok:
                bad += 1; print(f"cloud preset {preset} seed {seed}: stateless chain differs")
        print(f"clouds preset {preset}: {n_cloud} large irregular clouds compared")
print("mismatches:", bad)
sys.exit(1 if bad else 0)
