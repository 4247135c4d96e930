"""`python bench.py --gpus 2` with no launcher starts its own two ranks (host logic only: --dry-run-ms replaces the GPU
step by a sleep; rendezvous over gloo on 127.0.0.1).  The launcher path of the real bench is the same code."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + extra, env=env, capture_output=True, text=True,
                          timeout=240)


def test_self_launch_two_ranks():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run-ms", "20"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                      # rank 0 prints the one JSON line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak"
    assert len(rec["per_rank"]["frames_per_s"]) == 2
    assert rec["per_rank"]["min"] <= rec["per_rank"]["max"]
    # two ranks x 3 frames in >= 3 x 20 ms: the aggregate cannot exceed 2 / 20 ms
    assert 0 < rec["value"] <= 2 / 0.020 * 1.001


def test_world_size_mismatch_is_an_error():
    p = _run(["--gpus", "1", "--dry-run-ms", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_failed_rank_fails_the_launch():
    p = _run(["--gpus", "2", "--steps", "0", "--dry-run-ms", "1"])     # 0 steps: every rank divides by zero
    assert p.returncode != 0


def test_self_launch_eight_ranks():
    """the driver's 8-GPU shape: eight self-launched ranks rendezvous on 127.0.0.1, the barrier + max-over-
    ranks clock works, rank 0 prints ONE line whose per-rank report has eight entries"""
    p = _run(["--gpus", "8", "--steps", "4", "--warmup", "1", "--dry-run-ms", "25", "--frames-per-gpu", "4"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["steps"] == 4 and rec["scaling"] == "weak"
    assert len(rec["per_rank"]["frames_per_s"]) == 8
    # 8 ranks x 4 frames per step, steps of >= 25 ms: the aggregate cannot exceed 32 frames per 25 ms
    assert 0 < rec["value"] <= 32 / 0.025 * 1.001
    assert rec["per_rank"]["max"] <= 4 / 0.025 * 1.001


def test_roofline_object_from_a_profile_table():
    """bench.py's roofline / kernel_classes / frame_level objects, computed from a hand-made HIP-event table with
    round 3's kernel times: the dominant class is conv_pipe_kernel<64,256>, its binding roof is decided on the ISSUED matrix work
    (3 terms x flops > algorithmic bytes at the HBM peak -> "mfma", frac = issued fraction), the PMC bytes come from the committed
    summary, and the frame-level sums reproduce the review's figures (16.8-16.9 GB = 1.19-1.20 x algorithmic, 3.5 TB/s, 0.79 PFLOP/s,
    9 % overlap)."""
    import types
    sys.path.insert(0, REPO)
    import bench
    from vstnet_amd import _lib
    kid = _lib.kernel_id
    us = 1e-3
    table = {kid(16, 4, 1): (21.4 * us * 19, 19), kid(4, 16, 1): (28.7 * us * 19, 19), kid(16, 16, 2): (17.0 * us * 2, 2),
             kid(64, 16, 1): (22.7 * us * 18, 18), kid(16, 64, 1): (39.0 * us * 20, 20), kid(64, 64, 2): (33.0 * us * 2, 2),
             kid(256, 64, 1): (53.5 * us * 22, 22), kid(64, 64, 1): (18.3 * us * 24, 24), kid(64, 256, 1): (55.4 * us * 24, 24),
             1: (0.032, 1), 2: (0.023, 1), 5: (0.045, 2), 6: (0.022, 1), 7: (0.0595, 1)}
    args = types.SimpleNamespace(precision="bf16x3", effective_precision="bf16x3", mode="photo", recompute_style=False)
    roof, stages, classes, frame = bench.roofline_from_table(table, 1, 1, 1024, 1024, args, _lib, ms_per_step=4.839, frames_in_flight=3)
    assert "conv_pipe_kernel<64,256>" in roof["kernel"] and roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s"
    assert abs(roof["frac"] - 0.4186) < 2e-3 and abs(roof["hbm_frac"] - 0.3407) < 2e-3 and abs(roof["algorithmic_frac"] - 0.1395) < 1e-3
    assert roof["both_roofs"]["min_us_at_peak"] == {"hbm": 18.87, "mfma_issued": 23.19}
    assert roof["traffic"] and 1.0 < roof["traffic"] / roof["both_roofs"]["hbm"]["algorithmic_bytes"] < 1.1
    assert roof["traffic_source"]["file"].startswith("profiles/r04_") and roof["traffic_source"]["precision"] == "bf16x3"
    assert [c["bound"] for c in classes] == ["mfma", "mfma", "hbm", "hbm", "mfma"] and len(classes) == 5
    assert classes[2]["kernel"].startswith("conv_pair_kernel<16,64>") and 3.9e3 < classes[2]["hbm"]["pmc_GBps"] < 4.1e3
    assert frame["algorithmic_bytes_per_frame"] == 14118027264
    assert 1.18 < frame["pmc_over_algorithmic"] < 1.21 and 3.4 < frame["chip_average_pmc_TBps"] < 3.6
    assert 0.78 < frame["chip_average_issued_PFLOPps"] < 0.80 and 0.08 < frame["overlap_of_kernel_time"] < 0.10
    assert abs(stages["sum_ms_per_frame"] - sum(v["ms_per_frame"] for k, v in stages.items() if isinstance(v, dict))) < 1e-3
    # an HBM-bound dominant class reads "hbm": drop stage 3 from the table
    t2 = {k: v for k, v in table.items() if k < 65536 or (k >> 16) < 64 or ((k >> 4) & 0xFFF) < 64}
    roof2 = bench.roofline_from_table(t2, 1, 1, 1024, 1024, args, _lib, ms_per_step=2.0)[0]
    assert roof2["bound"] == "hbm" and roof2["unit"] == "GB/s" and "mfma_issued_frac" in roof2
