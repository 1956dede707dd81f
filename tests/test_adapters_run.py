"""include/mot_adapters.hpp's tracker functions RUN (emulator build of the kernels): a stream that creates far more tracks than
max_tracks_total keeps returning the reference's one-record-per-track-ever outputs, and outlives its max_tracks_ever budget."""
import os
import sys

import pytest

import adapter_case
import nodes_build as NB

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
pytestmark = pytest.mark.skipif(not NB.have_reference(), reason="the PCL shim needs the reference's vendored Eigen")


def test_adapter_tracker_outgrows_its_buffer_and_outlives_its_budget(oracle, tmp_path):
    import build_emu
    driver = NB.adapter_driver(build_emu.build())
    adapter_case.run(driver, oracle, tmp_path)
    adapter_case.run_refused(driver, tmp_path)
