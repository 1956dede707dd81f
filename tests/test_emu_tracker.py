"""CPU-only development check of the tracker kernel's LOGIC under tests/emu/hipemu.h against the golden vectors
(see hipemu.h: not a product path, not a parity claim)."""
import os
import sys

import numpy as np
import pytest

import golden_util as G

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.mark.parametrize("name", G.TRACKERS)
def test_emu_tracker_golden(mot, name):
    import build_emu
    lib = build_emu.build()
    fx = G.load(name)
    with mot.Context(lib_path=lib, max_points=4096, max_tracks_total=256) as c:
        for f in range(14):
            ts = 1.0e9 + f * float(fx["unit"])
            ego = c.ego_update(ts, 2.0 + 0.05 * f, 0.004 * f)
            assert np.allclose(ego, fx["ego"][f], rtol=1e-12, atol=1e-12)
            out = c.track_step(fx["boxes"][f][: fx["n_boxes"][f]], ts)
            G.check_tracker_frame(fx, f, out, lambda i: c.track_state(i), rtol=1e-6)
