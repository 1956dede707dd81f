"""The node shells (ros/src/*_node.cpp) — and the reference's own node sources on the adapter header — linked against libmot_hip.so, run on the MI355X, against the reference's own node
executables on the same message logs: every topic of `ground` and `cluster` byte for byte, the tracker's markers to 1e-4.
Both sets of executables are built where /root/reference exists (__graft_entry__.build() -> oracle/_ref/bin, ros/bin) and
travel to the GPU box prebuilt; without them the test is skipped. See tests/test_nodes.py for the mini-ROS they run on."""
import os

import pytest

import nodes_build as NB
import nodes_util as U

pytestmark = pytest.mark.gpu


def test_gpu_nodes_publish_what_the_reference_nodes_publish(hip_lib, synth, tmp_path):
    ref, own = NB.prebuilt(NB.REF_BIN), NB.prebuilt(NB.HIP_BIN)
    if ref is None or own is None:
        pytest.skip("prebuilt node executables (oracle/_ref/bin, ros/bin) are not on this box")
    os.makedirs(tmp_path / "ref"); os.makedirs(tmp_path / "own")
    chain = U.reference_chain(ref, synth, tmp_path / "ref")
    U.check_against_reference(own, chain, synth, tmp_path / "own")


def test_gpu_pipeline_node_publishes_what_ot0_main_publishes(hip_lib, synth, tmp_path):
    """the single-process node (one upload, stages chained on the resident cloud) against object_tracking0's own main.cpp"""
    ref, own = NB.prebuilt(NB.REF_BIN), NB.prebuilt(NB.HIP_BIN)
    if ref is None or own is None:
        pytest.skip("prebuilt node executables (oracle/_ref/bin, ros/bin) are not on this box")
    U.check_pipeline(ref["pipeline0"], own["pipeline"], synth, tmp_path)


def test_gpu_reference_node_sources_with_the_adapter_header(hip_lib, synth, tmp_path):
    """INTEGRATION.md's recipe on the MI355X: the reference's UNMODIFIED main.cpp's, their algorithm includes swapped for
    include/mot_adapters.hpp, linked against libmot_hip.so (ros/bin/recipe_*) — against the reference's own node executables."""
    ref, rec = NB.prebuilt(NB.REF_BIN), NB.prebuilt_recipe()
    if ref is None or rec is None:
        pytest.skip("prebuilt executables (oracle/_ref/bin, ros/bin/recipe_*) are not on this box")
    os.makedirs(tmp_path / "ref"); os.makedirs(tmp_path / "rec")
    chain = U.reference_chain(ref, synth, tmp_path / "ref")
    U.check_against_reference(rec, chain, synth, tmp_path / "rec")


def test_gpu_adapter_tracker_outgrows_its_buffer_and_outlives_its_budget(hip_lib, oracle, tmp_path):
    """the adapter header's immUkfJpdaf on libmot_hip.so with a small track budget (tests/adapter_case.py)"""
    rec = NB.prebuilt_recipe()
    if rec is None:
        pytest.skip("ros/bin/adapter_tracker_driver is not on this box")
    import adapter_case
    adapter_case.run(rec["adapter_tracker_driver"], oracle, tmp_path)
    adapter_case.run_refused(rec["adapter_tracker_driver"], tmp_path)
