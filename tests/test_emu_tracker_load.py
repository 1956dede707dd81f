"""tests/tracker_cases.py on the emulator build (CPU): 64 simultaneously live tracks per stream through mot_track_steps_dev
against the oracle — isolated gates and crowded (shared) gates — and angles far beyond the 32 turns up to which wrap_pi runs the
reference's loop. A check of the kernels' LOGIC; the -m gpu twin (tests/test_tracker_gpu.py) runs the same on the MI355X."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


def _host(a):
    a = np.ascontiguousarray(a)
    return a.ctypes.data, lambda keep=a: None


@pytest.mark.parametrize("spacing,min_live", [(9.0, 64), (2.0, 20)])
def test_64_live_tracks_vs_oracle_emulated(mot, oracle, spacing, min_live):
    import build_emu
    import tracker_cases as TC
    st = TC.many_live_tracks(mot, oracle, _host, lib_path=build_emu.build(), streams=2, T=64, frames=18, spacing=spacing, min_live=min_live)
    assert st["live_max"] >= min_live


def test_angles_beyond_32_turns_emulated(mot, oracle):
    import build_emu
    import tracker_cases as TC
    TC.angle_far_beyond_32_turns(mot, oracle, lib_path=build_emu.build())


def test_long_run_on_bounded_track_slots_emulated(mot, oracle):
    """far more tracks created than there are slots: every frame equal to the oracle with unbounded memory (eviction one step after death,
    positions of dead tracks kept for the merge step)"""
    import build_emu
    import tracker_cases as TC
    st = TC.long_run_bounded_slots(mot, oracle, lib_path=build_emu.build(), frames=700, slots=16, spots=9)
    assert st["tracks_ever"] >= 64
    if "reference_builds_stepped" in st:   # the head of the run also met the reference's own builds (narrow criterion + their noise floor)
        print(st["reference_builds_stepped"], st.get("reference_frames"), st.get("reference_builds_retired_at"), st.get("head_vs_reference_builds"))
        assert st.get("reference_frames", 0) >= 40 and st["head_vs_reference_builds"]["state_compares"] > 50
