"""CPU: `video_transfer.py --gpus N` (BASELINE config 5 end to end, SURVEY 8(e) "host gathers outputs") — the launcher, the
contiguous shards and the ordered merge, driven with world 2 on a stubbed stylise step (--stub_stylise: the "stylised" frame is
the resized input frame; no GPU is touched).  The real step differs only in what a child does with its frames."""
import os
import subprocess
import sys

import numpy as np
import pytest
from PIL import Image

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clip(d, n, sizes=None):
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(0)
    for i in range(n):
        h, w = (sizes or {}).get(i, (36, 52))
        img = rng.integers(0, 255, (h, w, 3), dtype=np.uint8)
        img[:4, :4] = i                                    # the frame's index, readable in the output
        Image.fromarray(img).save(os.path.join(d, "%03d.png" % i))


def _run(args, env_extra=None, timeout=120):
    env = dict(os.environ)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "video_transfer.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout, cwd=REPO)


def test_two_shards_merge_in_order(tmp_path):
    _clip(tmp_path / "clip", 7, sizes={3: (40, 60)})          # frame 3 has another size: resized on its own, no error
    Image.fromarray(np.zeros((20, 20, 3), np.uint8)).save(tmp_path / "s.png")
    p = _run(["--video", str(tmp_path / "clip"), "--style", str(tmp_path / "s.png"), "--out_dir", str(tmp_path / "o"),
              "--gpus", "2", "--stub_stylise", "--max_size", "48"])
    assert p.returncode == 0, p.stderr[-2000:]
    out_dir = tmp_path / "o" / "clip_s"
    names = sorted(os.listdir(out_dir))
    assert names == ["%05d.png" % i for i in range(7)]        # every frame once, contiguous shards 4 + 3 merged in order
    from utils.utils import img_resize
    from video_transfer import writer_size
    first = Image.open(tmp_path / "clip" / "000.png")
    wsz = writer_size(first, 48)
    for i, nme in enumerate(names):
        got = np.asarray(Image.open(out_dir / nme))
        assert got.shape[:2] == (wsz[1], wsz[0])              # one writer size for the clip (video_transfer.py:82-86)
        src = img_resize(Image.open(tmp_path / "clip" / ("%03d.png" % i)).convert("RGB"), 48, 4)
        want = np.asarray(src.resize(wsz, Image.BICUBIC))
        assert np.array_equal(got, want), i                   # the frame at index i IS frame i
    # the children were bound to one GPU each and capped in CPU threads
    from vstnet_amd.sharding import rank_environment, rank_threads
    e = rank_environment(1, 2, visible_device=True)
    assert e["HIP_VISIBLE_DEVICES"] == "1" and e["LOCAL_RANK"] == "0" and e["RANK"] == "1" and e["WORLD_SIZE"] == "2"
    assert int(e["OMP_NUM_THREADS"]) == rank_threads(2) >= 1


def test_failed_child_fails_the_run_and_incomplete_merge_is_an_error(tmp_path):
    _clip(tmp_path / "clip", 4)
    # a style file that does not exist: the children (not stubbed) fail before any GPU call -> non-zero exit, no merge
    p = _run(["--video", str(tmp_path / "clip"), "--style", str(tmp_path / "missing.png"), "--out_dir", str(tmp_path / "o"),
              "--gpus", "2", "--synthetic_weights"])
    assert p.returncode != 0
    from video_transfer import merge_outputs
    d = tmp_path / "frames"
    os.makedirs(d)
    for i in (0, 1, 3):
        Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(d / ("%05d.png" % i))
    with pytest.raises(RuntimeError, match="1 frames missing"):
        merge_outputs(str(d), 4, str(tmp_path), "x", 30, (8, 8))
