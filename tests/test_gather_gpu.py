"""The per-step exchange of live-track blocks (multi.TrackGather) on the MI355X over RCCL: a one-rank process group
with a receive buffer of its own — the export on the context's stream, the collective on torch's, ordered by stream waits only —
must deliver what mot_get_tracks returns after a synchronise. Run in a process of its own (it initialises torch.distributed);
the N > 1 form of the same code runs on CPU in tests/test_distributed_cpu.py (gloo, two ranks)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_track_gather_over_rccl_one_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_gather_gpu.py")], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "gather check ok" in r.stdout, (r.stdout[-800:], r.stderr[-1500:])
