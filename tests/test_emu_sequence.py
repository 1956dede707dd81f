"""tests/seq_parity.py (the end-to-end sequence checker of tests/test_sequence_gpu.py and bench.py's parity_check) exercised on
the emulator build of the kernels — a check of the CHECKER's logic and of the fused path's host side on CPU, not a parity claim
(tests/emu/hipemu.h). Two ragged streams with different timestamp units, a moving ego, every frame compared."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


import pytest


@pytest.mark.parametrize("which", ["restatement", "reference build first"])
def test_sequence_checker_on_the_emulator(mot, synth, oracle, which):
    """which: the oracle the checker is given — the restatement, or (as on the GPU box) oracle_lib.RefFirst"""
    import build_emu
    import seq_parity as SP
    lib = build_emu.build()
    if which != "restatement":
        if oracle.ref() is None:
            pytest.skip("oracle/_ref is not on this box")
        oracle = oracle.RefFirst(oracle)
    B, N, stride, F = 2, 6000, 6144, 8
    clouds = np.zeros((F, B, stride, 4), np.float32)
    n_seq = np.zeros((F, B), np.int32)
    for f in range(F):
        for b in range(B):
            n = N - 500 * b - 7 * f
            clouds[f, b, :n] = synth.make_cloud(N, 20 + b, f)[:n]
            n_seq[f, b] = n
    ego_v = 2.0 + 0.2 * np.arange(F); ego_yaw = 0.01 * np.arange(F)
    p = oracle.params(0)
    with mot.Context(lib_path=lib, max_points=stride, max_batch=B, max_tracks_total=256) as c:
        st = SP.check_sequence(c, oracle, p, lambda f: clouds[f].ctypes.data, lambda f, b: clouds[f, b], n_seq, stride, ego_v, ego_yaw, units=[1e5, 0.1],
                               skip_ill_conditioned=True, noise_floor=True, mar_check=True)   # the GPU run's settings (tests/seq_parity_gpu_run.py)
    assert st["frames"] == F and st["boxes"] > 0 and st["tracks_ever"] > 0 and st["max_rel_state_err"] <= SP.RTOL
    assert st["above_1e-4_unexplained"] == 0 and st["mar_clusters_cross_checked"] > 0
    assert st["tracker_oracle"].startswith("restatement" if which == "restatement" else "reference build")
    if which != "restatement":
        assert ("box_fit", "reference build (box -> cluster ids: restatement)") in oracle.used and ("Tracker", "reference build") in oracle.used
