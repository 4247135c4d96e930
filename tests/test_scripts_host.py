"""CPU: host helpers of the drop-in scripts (utils/utils.py, video_transfer.writer_size) against the oracle's
restatement of the reference's loops."""
import numpy as np
from PIL import Image

from oracle import cpu_ref
from utils.utils import img_resize, colors_to_labels, load_segment


def test_img_resize_floor_to_multiple_of_4():
    img = Image.fromarray(np.random.default_rng(0).integers(0, 255, (675, 1200, 3), dtype=np.uint8))
    out = img_resize(img, 1280, down_scale=4)
    assert out.size == (1200, 672)                       # reference comment: content [1,3,672,1200] for 05.jpg
    big = Image.fromarray(np.zeros((2400, 3840, 3), np.uint8))
    assert img_resize(big, 1280, down_scale=4).size == (1280, 800)
    assert img_resize(big, 1280).size == (1280, 800)
    assert img_resize(Image.fromarray(np.zeros((30, 50, 3), np.uint8)), 1280, 4).size == (48, 28)


def test_colors_to_labels_matches_reference_loop(tmp_path):
    rng = np.random.default_rng(1)
    exact = np.array([[0, 0, 255], [0, 255, 0], [0, 0, 0], [255, 255, 255], [255, 0, 0], [255, 255, 0], [128, 128, 128],
                      [0, 255, 255], [255, 0, 255]], dtype=np.uint8)
    img = exact[rng.integers(0, 9, (12, 17))]
    noisy = rng.integers(0, 256, (12, 17, 3), dtype=np.uint8)
    img[::3, ::2] = noisy[::3, ::2]                      # off-palette pixels -> nearest colour
    img[0, 0] = (128, 0, 0)                              # equidistant from (0,0,0) and (255,0,0)? 128 vs 127 -> black... and a true tie:
    img[0, 1] = (64, 64, 64)                             # 192 from black, 192 from grey -> tie keeps the earlier key (black)
    ref = cpu_ref.colors_to_labels_loop(img)
    assert np.array_equal(colors_to_labels(img), ref)
    assert colors_to_labels(img)[0, 1] == 0
    p = tmp_path / "seg.png"
    Image.fromarray(img).save(p)
    assert np.array_equal(load_segment(str(p)), ref)
    assert load_segment(str(p), size=(34, 24)).shape == (24, 34)
    assert load_segment(str(tmp_path / "missing.png")) is None


def test_writer_size_quirk():
    from video_transfer import writer_size
    frame = Image.fromarray(np.zeros((1080, 1920, 3), np.uint8))
    assert writer_size(frame, 1280) == (1280, 1080)      # video_transfer.py:83-86: width is overwritten first
    assert writer_size(frame, 1920) == (1920, 1080)


def test_seg_remapping_matches_reference_loops():
    from models.segmentation.SegReMapping import SegReMapping
    rng = np.random.default_rng(7)
    K = 20
    mapping = np.stack([rng.permutation(K) for _ in range(K)], axis=1)      # column l: related labels, best first
    mapping[-1] = np.arange(K)                                               # the reference's table ends with identity
    fast, slow = SegReMapping(mapping, min_ratio=0.02), cpu_ref.SegReMappingLoop(mapping, min_ratio=0.02)
    for trial in range(5):
        seg = rng.choice(K, size=(40, 56), p=rng.dirichlet(np.full(K, 0.3))).astype(np.uint8)
        seg[0, :3] = (trial + 3) % K                                         # a tiny region -> remapped
        sty = rng.choice(K, size=(30, 30), p=rng.dirichlet(np.full(K, 0.2))).astype(np.uint8)
        a, b = fast.self_remapping(seg), slow.self_remapping(seg)
        assert a.dtype == seg.dtype and np.array_equal(a, b)
        assert np.array_equal(fast.cross_remapping(a, sty), slow.cross_remapping(b, sty))
        assert np.array_equal(seg, seg.copy())
