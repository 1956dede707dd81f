"""The frame-per-workgroup compaction kernel (ground.hip, large batches) on the emulator build — development check of the
kernel's LOGIC on CPU (chunk boundaries, the ragged last chunk, empty and one-point frames, both presets, the crop); the parity
claim is tests/test_frame_kernel_gpu.py on the MI355X."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import frame_kernel_case


@pytest.fixture(scope="module")
def emu_lib():
    import build_emu
    return build_emu.build()


@pytest.mark.parametrize("sizes,stride,preset,crop", [
    ([9000, 1, 0, 4096, 4097, 8191, 12288, 5000], 12288, 0, False),
    ([7000, 4095, 3000], 8192, 1, False),
    ([9000, 6000], 9216, 0, True),
])
def test_frame_kernel_equals_chunk_kernel_and_oracle(mot, oracle, synth, emu_lib, sizes, stride, preset, crop):
    frame_kernel_case.run(mot, oracle, synth, emu_lib, sizes, stride, preset=preset, frames=2, crop=crop)


def test_frame_kernel_on_irregular_frames(mot, oracle, synth, emu_lib):
    """several hundred tiny clusters (the occupancy planes it leaves are speckled), and a frame with more than 65536 elevated
    points (many chunks, long running positions), next to normal frames in the same batch"""
    p = oracle.params(0)
    many = frame_kernel_case.many_clusters_cloud(); lifted = frame_kernel_case.crowded_cloud(oracle, synth, 60000, 9); normal = synth.make_cloud(9000, 4, 0)
    g = oracle.ground_remove(p, many)
    assert oracle.cluster(p, g["elevated"])["num_cluster"] > 255
    assert len(oracle.ground_remove(p, lifted)["elevated"]) > 65536
    clouds = [many, lifted, normal]
    frame_kernel_case.run(mot, oracle, synth, emu_lib, [len(x) for x in clouds], max(len(x) for x in clouds) + 64, frames=1, clouds_override=clouds)
