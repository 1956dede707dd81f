"""Static resource usage of the product's kernels (hipcc -Rpass-analysis=kernel-resource-usage, no GPU): a kernel that starts to spill, or
loses the occupancy its design counts on, shows here before it shows in a profile. (Round 4: a second caller stopped the compiler inlining
process_imm_ukf and the four-launch prediction kernel silently got a stack frame — 12 bytes of scratch per lane.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_no_kernel_spills_and_the_streaming_kernels_keep_their_occupancy():
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc on this box")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")], capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = {}
    for line in r.stdout.splitlines()[1:]:
        f = line.split()
        if len(f) >= 8:
            rows[f[1]] = dict(file=f[0], vgpr=int(f[2]), scratch=int(f[5]), waves=int(f[6]), lds=int(f[7]))
    for k in ("polar_minz_kernel", "classify_compact_elevated_kernel", "ccl_kernel", "label_stats_kernel", "cluster_gather_kernel", "track_predict_kernel",
              "track_update_kernel", "track_finish_kernel", "track_step_stream_kernel", "box_markers_kernel", "side_scatter_kernel"):
        assert k in rows, (k, sorted(rows))
    # (track_update_dense_kernel is the ONE kernel that spills, by design: the same code as track_update_kernel held to 3 waves per SIMD for launches
    # with more than 24 k live tracks, chosen on the device — profiles/r06_tracker_occupancy_variants.txt)
    spilled = {k: v["scratch"] for k, v in rows.items() if v["scratch"] != 0 and k != "track_update_dense_kernel"}
    assert not spilled, spilled
    assert rows["track_update_dense_kernel"]["waves"] >= 3 and rows["track_update_dense_kernel"]["scratch"] <= 256, rows["track_update_dense_kernel"]
    # the streaming kernels are latency hiders: 8 (7) waves per SIMD is what their block sizes and LDS budgets assume
    # (label_stats_kernel: 256 threads x 8 points since round 5, LDS-bound, each wave with twice the loads in flight: 151 against 155-159 us per
    # 512 frames, profiles/r05_label_geometry_ab.txt; its per-tile counts and slot numbers packed into shorts / bytes put a SIXTH workgroup on
    # a CU: +3 % on the four-context line, profiles/r05_label_lds_ab.txt)
    for k, w in (("polar_minz_kernel", 8), ("label_stats_kernel", 6), ("classify_compact_elevated_kernel", 7), ("ccl_kernel", 8), ("cluster_index_kernel", 8)):
        assert rows[k]["waves"] >= w, (k, rows[k])
    # the tracker's register budgets (launch bounds): three / two waves per SIMD for prediction / update, the one-launch step inside 160 KB of LDS
    assert rows["track_predict_kernel"]["waves"] >= 3 and rows["track_update_kernel"]["waves"] >= 2
    assert rows["track_step_stream_kernel"]["waves"] >= 2 and rows["track_step_stream_kernel"]["lds"] <= 160 * 1024
