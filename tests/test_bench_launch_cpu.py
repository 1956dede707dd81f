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
