"""bench.py's launch logic on CPU: `python bench.py --gpus N` with no launcher around it must start N ranks itself
(torch.distributed.run, 127.0.0.1), rank 0 prints ONE JSON line with n_gpus = N. Runs the script's --selftest-cpu mode: gloo,
the emulator build of the kernels, tiny clouds — the numbers are not measurements, the line says so."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "1", "--selftest-cpu"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [1, 2])
def test_bench_self_spawns_n_ranks(n):
    pytest.importorskip("torch")
    out = _run(n)
    assert out["n_gpus"] == n and out["steps"] == 1 and out["warmup"] == 1
    assert out["config"]["streams"] == 2 * n and out["scaling"] == "weak" and out["value"] > 0
    assert "SELFTEST" in out["data"]
    for key in ("metric", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "config"):
        assert key in out


def test_bench_refuses_a_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-cpu"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_bench_parity_check_logic_on_the_emulator():
    """bench.py's `parity_check` (the GPU's per-frame results of bench stream 0 against what the reference computes in the
    cpu_baseline leg) with the emulator build standing in for the GPU and tiny frames: the plumbing, and that a corrupted result is
    reported — the real comparison runs inside `python bench.py` on the MI355X."""
    torch = pytest.importorskip("torch")
    import importlib.util
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from conftest import load_pkg, load_sub
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    mot = load_pkg(); synth = load_sub("synth")
    lib = build_emu.build()
    F, N, stride = 6, 6000, 6144
    host = np.zeros((F, 1, stride, 4), np.float32)
    n_seq = np.zeros((F, 1), np.int32)
    for f in range(F):
        n = N - 11 * f
        host[f, 0, :n] = synth.make_cloud(N, 4, f)[:n]; n_seq[f, 0] = n
    seq = torch.from_numpy(host)
    ego_v = np.full(F, 2.0); ego_yaw = 0.01 * np.arange(F)
    res = bench.gpu_sequence_results(mot, 0, seq, n_seq, stride, ego_v, ego_yaw, 0, lib_path=lib)
    base, par = bench.cpu_baseline(host[:, 0, :N], ego_v, ego_yaw, N, budget_s=2.0, gpu_results=res, n_per_frame=n_seq[:, 0], lib=mot.load_library(lib), quick=True)
    assert base["value"] > 0 and par["frames"] == F
    assert par["masks_boxes_bit_exact"] and par["track_sets_equal"] and par["boxes_total"] > 0 and par["first_mismatch_frame"] is None, par
    assert par["max_rel_state_err"] is not None and par["max_rel_state_err"] <= 1e-4
    assert par["states_within_1e-4"] and par["states_within_bar"] and not par["states_explained_by_reference_noise"]
    assert par["track_frames_above_1e-4"] == 0 and par["track_frames_above_1e-4_unexplained"] == 0
    assert {"restatement", "novec"} <= set(par["noise_floor_replicas"]["in_use_last_frame"]) and par["min_area_rect_cross_check"]["failed"] == 0
    # a state off by 1e-3 on a perfectly conditioned track: above the bar, NOT explained by the reference's own noise -> the flag must fall
    import copy
    res_bad = copy.deepcopy(res)
    i = sorted(res_bad[5]["states"])[0]
    res_bad[5]["states"][i]["x_merge"] = res_bad[5]["states"][i]["x_merge"] * (1 + 1e-3)
    _, par3 = bench.cpu_baseline(host[:, 0, :N], ego_v, ego_yaw, N, budget_s=2.0, gpu_results=res_bad, n_per_frame=n_seq[:, 0], lib=mot.load_library(lib), quick=True)
    assert not par3["states_within_1e-4"] and not par3["states_within_bar"] and par3["track_frames_above_1e-4"] == 1 and par3["track_frames_above_1e-4_unexplained"] == 1, par3
    res[3]["boxes"] = res[3]["boxes"] + np.float32(1e-3)   # a corrupted GPU result must show up, with its frame
    _, par2 = bench.cpu_baseline(host[:, 0, :N], ego_v, ego_yaw, N, budget_s=2.0, gpu_results=res, n_per_frame=n_seq[:, 0], lib=mot.load_library(lib), quick=True)
    assert not par2["boxes_bit_exact"] and not par2["masks_boxes_bit_exact"] and par2["first_mismatch_frame"]["boxes_bit_exact"] == 3


def test_stage_wise_leg_on_the_emulator(mot, synth):
    """bench.py's single_stream.stage_wise_host_buffers — the three nodes' call sequence on host buffers — run on the emulator build with a
    few small frames: the leg's code (C calls with NULL outputs, the change of frame, the per-call clocks) must work before a GPU sees it"""
    import functools
    import types
    torch = pytest.importorskip("torch")
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    sys.path.insert(0, ROOT)
    import build_emu
    import bench
    lib = build_emu.build()
    seqmod = bench._load("mot_amd.sequence", os.path.join(bench.PKG_DIR, "sequence.py"))
    F, N, stride = 4, 12000, 12288
    seq = np.zeros((F, 1, stride, 4), np.float32)
    for f in range(F):
        seq[f, 0, :N] = synth.make_cloud(N, 3, f)
    m = types.SimpleNamespace(**{k: getattr(mot, k) for k in dir(mot) if not k.startswith("__")})
    m.Context = functools.partial(mot.Context, lib_path=lib)
    r = bench.stage_wise_host_buffers(m, 0, torch.from_numpy(seq), np.full((F, 1), N, np.int32), stride, np.linspace(2, 3, F), np.linspace(0, 0.05, F), seqmod)
    assert r["frames"] == F and r["ms_per_frame"]["median"] > 0 and set(r["stage_ms"]) == {"ground", "cluster_box", "tracker"}
    assert set(r["cluster_box_calls_ms_median"]) == {"mot_cluster", "mot_cluster_products", "mot_box_fit_resident", "mot_box_markers"}
    assert r["tracks_ever"] >= 1
    cb = r["call_by_call"]   # both forms ran the same frames: same boxes, same tracks
    assert cb["boxes_last_frame"] == r["boxes_last_frame"] and cb["tracks_ever"] == r["tracks_ever"] and cb["ms_per_frame"]["median"] > 0
