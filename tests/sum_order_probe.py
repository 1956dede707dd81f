"""Run by tests/test_emu_tracker.py::test_sum_order_knob in a process of its own (MOT_EMU_DEFINES is read when build_emu is imported):
the golden tracker fixtures through the emulated kernels; prints the worst relative error of the live tracks' merged state / covariance /
mode probabilities against the reference build's values (tests/golden), one line per fixture."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, "emu"))
import build_emu      # noqa: E402
import conftest       # noqa: E402
import golden_util as G   # noqa: E402

mot = conftest.load_pkg()
lib = build_emu.build()
for name in G.TRACKERS:
    fx = G.load(name)
    worst = 0.0
    with mot.Context(mot.params(0, lib=mot.load_library(lib)), lib_path=lib, max_points=4096, max_tracks_total=256) as c:
        for f in range(14):
            ts = 1.0e9 + f * float(fx["unit"])
            c.ego_update(ts, *G.ego_of(fx, f))
            out = c.track_step(fx["boxes"][f][: fx["n_boxes"][f]], ts)
            n = int(fx["n_tracks"][f])
            assert out["n"] == n and np.array_equal(out["track_manage"], fx["track_manage"][f][:n])
            for i in np.nonzero(fx["track_manage"][f][:n] > 0)[0]:
                s = c.track_state(int(i))
                for k in ("x_merge", "p_merge", "mode_prob"):
                    ref = fx[k][f][i]
                    if np.all(np.isfinite(ref)):
                        worst = max(worst, float(np.abs(np.asarray(s[k]) - ref).max() / max(np.abs(ref).max(), 1e-300)))
    print(name, "%.3e" % worst)
