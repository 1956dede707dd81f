"""The frame-per-workgroup compaction kernel (ground.hip: the fused path's kernel from 384 frames per launch upwards) on the
MI355X, forced on at small batches: bit-identical clouds / mask / counts / cluster grid / boxes / tracks to the
chunk-per-workgroup kernel and to the oracle, at full frame sizes, chunk-boundary sizes, empty frames, both presets, with the
node's crop; and once at the batch size that selects it by itself."""
import numpy as np
import pytest

import frame_kernel_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sizes,stride,preset,crop", [
    ([120000, 1, 0, 4096, 4097, 99999, 200000, 131072], 200704, 0, False),
    ([120000, 65536, 77777], 120832, 1, False),
    ([120000, 100000], 120832, 0, True),
])
def test_frame_kernel_equals_chunk_kernel_and_oracle(mot, hip_lib, oracle, synth, sizes, stride, preset, crop):
    frame_kernel_case.run(mot, oracle, synth, None, sizes, stride, preset=preset, frames=2, crop=crop)


def test_frame_kernel_on_irregular_frames(mot, hip_lib, oracle, synth):
    """several hundred tiny clusters (the occupancy planes it leaves are speckled), and a frame with more than 65536 elevated
    points (many chunks, long running positions), next to normal frames in the same batch"""
    p = oracle.params(0)
    many = frame_kernel_case.many_clusters_cloud(); lifted = frame_kernel_case.crowded_cloud(oracle, synth, 120000, 4); normal = synth.make_cloud(120000, 4, 0)
    assert oracle.cluster(p, oracle.ground_remove(p, many)["elevated"])["num_cluster"] > 255
    assert len(oracle.ground_remove(p, lifted)["elevated"]) > 65536
    clouds = [normal, many, lifted, normal[::-1].copy()]
    frame_kernel_case.run(mot, oracle, synth, None, [len(x) for x in clouds], ((max(len(x) for x in clouds) + 2047) // 2048) * 2048, frames=2, clouds_override=clouds)


def test_frame_kernel_selected_by_batch_size(mot, hip_lib, oracle, synth):
    """400 small frames in one launch (>= 384: the frame kernel by default) against 400 frames through a context pinned to the
    chunk kernel"""
    import hiprt
    B, N, stride = 400, 6000, 6144
    host = np.zeros((B, stride, 4), np.float32)
    base = [synth.make_cloud(N, 50 + s, 0) for s in range(8)]
    sizes = [N - 13 * (s % 50) for s in range(B)]
    for s in range(B):
        host[s, : sizes[s]] = base[s % 8][: sizes[s]]
    dev = hiprt.DeviceBuffer(host)
    with mot.Context(max_points=stride, max_batch=B) as a, mot.Context(max_points=stride, max_batch=B) as b:
        assert b.lib.mot_debug_option(b._h, 0, 0) == 0
        a.frames_dev(dev.ptr, stride * 4, sizes); b.frames_dev(dev.ptr, stride * 4, sizes)
        a.synchronize(); b.synchronize()
        p = oracle.params(0)
        for s in list(range(0, B, 37)) + [B - 1]:
            ra, rb = a.get_ground(s, n_hint=N), b.get_ground(s, n_hint=N)
            assert np.array_equal(ra["elevated"], rb["elevated"]) and np.array_equal(ra["ground"], rb["ground"]) and np.array_equal(ra["mask"][: sizes[s]], rb["mask"][: sizes[s]])
            ca, cb = a.get_clusters(s), b.get_clusters(s)
            assert ca["num_cluster"] == cb["num_cluster"] and np.array_equal(ca["grid"], cb["grid"])
            assert np.array_equal(a.get_boxes(s)["boxes"], b.get_boxes(s)["boxes"])
            g = oracle.ground_remove(p, host[s, : sizes[s]])
            assert np.array_equal(ra["elevated"], g["elevated"])
    dev.free()
