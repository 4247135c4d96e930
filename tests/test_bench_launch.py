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
