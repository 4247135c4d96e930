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


def test_world_8_shards_and_ordered_merge(tmp_path):
    """the shape the driver's 8-GPU node runs: eight children, shard sizes as balanced as the frame count
    allows, the merged directory holds every frame once and in order; stale numbered frames of an earlier, longer run in the
    output directory are cleared before the children start, other files there are left alone."""
    from vstnet_amd.sharding import shard_range
    sizes = [shard_range(300, r, 8) for r in range(8)]
    assert [hi - lo for lo, hi in sizes] == [38, 38, 38, 38, 37, 37, 37, 37]          # config 5: 300 frames on 8 GPUs
    assert sizes[0][0] == 0 and sizes[-1][1] == 300 and all(a[1] == b[0] for a, b in zip(sizes, sizes[1:]))
    _clip(tmp_path / "clip", 19)
    Image.fromarray(np.zeros((20, 20, 3), np.uint8)).save(tmp_path / "s.png")
    out_dir = tmp_path / "o" / "clip_s"
    os.makedirs(out_dir)
    for i in (18, 19, 25):                                    # an earlier run of a longer clip left these behind
        Image.fromarray(np.full((8, 8, 3), 255, np.uint8)).save(out_dir / ("%05d.png" % i))
    (out_dir / "notes.txt").write_text("mine")
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(out_dir / "poster.png")
    p = _run(["--video", str(tmp_path / "clip"), "--style", str(tmp_path / "s.png"), "--out_dir", str(tmp_path / "o"),
              "--gpus", "8", "--stub_stylise", "--max_size", "48", "--workers", "2"], timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    names = sorted(f for f in os.listdir(out_dir) if f[0].isdigit())
    assert names == ["%05d.png" % i for i in range(19)]
    assert (out_dir / "notes.txt").read_text() == "mine" and (out_dir / "poster.png").exists()
    for i, nme in enumerate(names):
        got = np.asarray(Image.open(out_dir / nme))
        assert int(got[0, 0, 0]) == i or got.shape[0] != 36      # the frame's index survives where no resize happened
    assert [shard_range(19, r, 8) for r in range(8)] == [(0, 3), (3, 6), (6, 9), (9, 11), (11, 13), (13, 15), (15, 17), (17, 19)]


def test_abbreviated_flags_are_rejected_and_children_never_relaunch(tmp_path):
    """`--gpu 2` used to parse as --gpus 2, survive the child argv rewrite and make every child start its
    own children.  The parser takes no abbreviations now, and a child's command line ends in an explicit `--gpus 1`."""
    _clip(tmp_path / "clip", 3)
    Image.fromarray(np.zeros((20, 20, 3), np.uint8)).save(tmp_path / "s.png")
    p = _run(["--video", str(tmp_path / "clip"), "--style", str(tmp_path / "s.png"), "--out_dir", str(tmp_path / "o"),
              "--gpu", "2", "--stub_stylise"])
    assert p.returncode == 2 and "unrecognized arguments" in p.stderr
    p = _run(["--video", str(tmp_path / "clip"), "--style", str(tmp_path / "s.png"), "--out_dir", str(tmp_path / "o"),
              "--shar", "0/1", "--stub_stylise"])
    assert p.returncode == 2
    import video_transfer
    seen = {}

    def fake_launch(cmds, envs):
        seen["cmds"], seen["envs"] = cmds, envs
        return 0
    import vstnet_amd.sharding as sh
    real = sh.launch_children
    sh.launch_children = fake_launch
    try:
        argv = ["--video", str(tmp_path / "clip"), "--style", str(tmp_path / "s.png"), "--gpus=2", "--shard=0/1", "--stub_stylise"]
        args = video_transfer.build_parser().parse_args(argv)
        assert video_transfer.launch_shards(args, argv) == 0
    finally:
        sh.launch_children = real
    for r, cmd in enumerate(seen["cmds"]):
        assert cmd[-2:] == ["--gpus", "1"] and "--gpus=2" not in cmd and "--shard=0/1" not in cmd
        assert cmd[cmd.index("--shard") + 1] == "%d/2" % r
        assert video_transfer.build_parser().parse_args(cmd[2:]).gpus == 1


def test_children_honour_the_parents_visible_devices(tmp_path, monkeypatch):
    """a parent confined to HIP_VISIBLE_DEVICES=4,5,6,7 gives child r ITS r-th device (4, 5, ...), not physical GPU
    r; GPUs are counted without the HIP runtime (the restriction, else the KFD topology)."""
    from vstnet_amd.sharding import rank_environment, count_gpus, visible_device_list
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    assert rank_environment(3, 8, visible_device=True)["HIP_VISIBLE_DEVICES"] == "3"
    assert rank_environment(3, 8, visible_device=True, n_devices=2)["HIP_VISIBLE_DEVICES"] == "1"
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5,6,7")
    assert [rank_environment(r, 4, visible_device=True)["HIP_VISIBLE_DEVICES"] for r in range(4)] == ["4", "5", "6", "7"]
    assert rank_environment(5, 8, visible_device=True)["HIP_VISIBLE_DEVICES"] == "5"          # 8 ranks on 4 devices: r % 4
    assert count_gpus() == 4
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "2,3")
    e = rank_environment(1, 2, visible_device=True)
    assert e["ROCR_VISIBLE_DEVICES"] == "3" and "HIP_VISIBLE_DEVICES" not in e and visible_device_list()[1] == ["2", "3"]
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    # the KFD topology: two CPU nodes (no SIMDs) and three GPU nodes
    root = tmp_path / "nodes"
    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):
        os.makedirs(root / str(i))
        (root / str(i) / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\n")
    assert count_gpus(kfd_root=str(root)) == 3
    assert count_gpus(kfd_root=str(tmp_path / "absent")) == 0


def test_parallel_map_keeps_order_and_async_sink_uses_workers():
    import threading
    import time
    from vstnet_amd.pipeline import parallel_map, AsyncSink

    def slow(i):
        time.sleep(0.002 * ((i * 7) % 5))
        if i == 37:
            raise ValueError("frame 37")
        return i * i
    assert list(parallel_map(lambda i: i * i, range(50), workers=4, ahead=3)) == [i * i for i in range(50)]
    out = []
    with pytest.raises(ValueError, match="frame 37"):
        for v in parallel_map(slow, range(50), workers=4, ahead=3):
            out.append(v)
    assert out == [i * i for i in range(37)]                   # everything before the failing item arrived, in order
    seen, names, lock = [], set(), threading.Lock()

    def fn(i, frame):
        time.sleep(0.01)
        with lock:
            seen.append((i, int(frame[0, 0, 0])))
            names.add(threading.current_thread().name)
    sink = AsyncSink(fn, workers=4)
    buf = np.zeros((4, 4, 3), np.uint8)
    for i in range(24):
        buf[...] = i                                           # the slot is reused: the sink must have copied it
        sink(i, buf)
    sink.close()
    assert sorted(seen) == [(i, i) for i in range(24)] and len(names) > 1


def test_stored_png_writer_is_a_valid_lossless_png(tmp_path):
    """--png_level 0: the pipeline's own PNG writer (stored deflate blocks, no row filters) reads back bit for bit with PIL, for
    sizes whose rows are not multiples of anything; level > 0 goes through PIL"""
    from vstnet_amd.pipeline import save_png
    rng = np.random.default_rng(3)
    for h, w in ((1, 1), (7, 13), (270, 481), (1080, 1920)):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for level in (0, 1):
            path = tmp_path / f"{h}x{w}_{level}.png"
            save_png(str(path), a, level)
            img = Image.open(path)
            assert img.mode == "RGB" and img.size == (w, h) and np.array_equal(np.asarray(img), a)
        assert os.path.getsize(tmp_path / f"{h}x{w}_0.png") <= 3 * h * w + h + 200 + 5 * (1 + (3 * h * w + h) // 65535)
