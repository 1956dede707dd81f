"""The node shells (ros/src/*_node.cpp) linked against libmot_hip.so, run on the MI355X, against the reference's own node
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
