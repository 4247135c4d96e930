"""End-to-end (host-inclusive) frames/s of video_transfer.py on one GPU: files in -> numbered PNGs out (BASELINE config 5's
per-GPU work, video_transfer.py:160-214 of the reference: decode, resize, stylise with per-region cWCT, write).

    python tools/video_e2e.py [--frames 120] [--height 1080 --width 1920] [--masked 5] [--workers 0] [--png-level 1]

Builds a synthetic clip in a temporary directory (smooth, natural-like frames as JPEG - what a decoded video looks like to the
loop -, one style image, 5-band colour-coded segmentation maps), runs `video_transfer.main` on it and prints one JSON line:
the script's frames/s, and next to it the rates of the host stages alone (decode + resize, PNG encode) per worker thread.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from utils.utils import SEG_COLORS                              # noqa: E402


def natural_frame(h, w, t, seed=0):
    """low-frequency colour fields + a little texture: compresses like a photograph, not like noise"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 3), np.float32)
    for c in range(3):
        for k in range(4):
            fx, fy, ph = rng.uniform(0.5, 6) / w, rng.uniform(0.5, 6) / h, rng.uniform(0, 6.28)
            img[..., c] += np.sin(6.28 * (fx * x + fy * y) + ph + 0.05 * t * (k + 1)) / (k + 1)
    img = (img - img.min()) / (img.max() - img.min())
    img += rng.normal(0, 0.02, img.shape).astype(np.float32)
    return (np.clip(img, 0, 1) * 255).astype(np.uint8)


def band_mask(h, w, k):
    out = np.zeros((h, w, 3), np.uint8)
    edges = np.linspace(0, w, k + 1).astype(int)
    for i in range(k):
        out[:, edges[i]:edges[i + 1]] = SEG_COLORS[i][0]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--masked", type=int, default=5)
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--png-level", type=int, default=0)
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--precision", default=None)
    ap.add_argument("--gpus", type=int, default=1)
    args = ap.parse_args()
    H, W = args.height, args.width
    with tempfile.TemporaryDirectory(prefix="vst_e2e_") as d:
        clip = os.path.join(d, "clip")
        os.makedirs(clip)
        base = [natural_frame(H, W, t, seed=7) for t in range(8)]           # 8 distinct frames, cycled
        for i in range(args.frames):
            Image.fromarray(base[i % 8]).save(os.path.join(clip, "%04d.jpg" % i), quality=92)
        Image.fromarray(natural_frame(H, W, 3, seed=11)).save(os.path.join(d, "style.jpg"), quality=92)
        argv = ["--video", clip, "--style", os.path.join(d, "style.jpg"), "--out_dir", os.path.join(d, "out"), "--max_size",
                str(max(H, W)), "--synthetic_weights", "--streams", str(args.streams), "--workers", str(args.workers),
                "--png_level", str(args.png_level), "--depth", "6"]
        if args.precision:
            argv += ["--precision", args.precision]
        if args.masked:
            Image.fromarray(band_mask(H, W, args.masked)).save(os.path.join(d, "cseg.png"))
            Image.fromarray(band_mask(H, W, args.masked)[:, ::-1].copy()).save(os.path.join(d, "sseg.png"))
            argv += ["--content_seg", os.path.join(d, "cseg.png"), "--style_seg", os.path.join(d, "sseg.png")]
        if args.gpus > 1:
            argv += ["--gpus", str(args.gpus)]
        # host stages alone, one thread: what a worker does per frame
        from utils.utils import img_resize
        from vstnet_amd.pipeline import save_png
        t0 = time.perf_counter()
        for i in range(8):
            arr = np.asarray(img_resize(Image.open(os.path.join(clip, "%04d.jpg" % i)).convert("RGB"), max(H, W), 4), dtype=np.uint8)
        t_dec = (time.perf_counter() - t0) / 8
        t0 = time.perf_counter()
        for i in range(8):
            save_png(os.path.join(d, "enc_%d.png" % i), arr, args.png_level)
        t_enc = (time.perf_counter() - t0) / 8
        png_mb = os.path.getsize(os.path.join(d, "enc_0.png")) / 1e6
        import video_transfer
        from vstnet_amd.pipeline import host_workers
        # (the first run also builds / loads the network: time a second one)
        small = argv[:]
        t0 = time.perf_counter()
        video_transfer.main(argv)
        t_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        out = video_transfer.main(argv)
        dt = time.perf_counter() - t0
        n_out = len([f for f in os.listdir(out) if f.endswith(".png")])
        assert n_out == args.frames, (n_out, args.frames)
        # the same on the first quarter of the clip: the difference is the rate of the frame loop without the per-run setup
        # (network build + weight packing, style encode, mask plan, ring buffers)
        q = os.path.join(d, "clipq")
        os.makedirs(q)
        nq = max(8, args.frames // 4)
        for i in range(nq):
            os.symlink(os.path.join(clip, "%04d.jpg" % i), os.path.join(q, "%04d.jpg" % i))
        argq = [q if a == clip else a for a in argv]
        t0 = time.perf_counter()
        video_transfer.main(argq)
        dq = time.perf_counter() - t0
        rec = {"what": "video_transfer.py end to end on one GPU: JPEG frames in, numbered PNGs out, host-inclusive",
               "frames": args.frames, "size": f"{W}x{H}", "masked_labels": args.masked, "gpus": args.gpus,
               "precision": args.precision or "bf16x3 (default)", "frames_per_s": round(args.frames / dt, 2),
               "seconds": round(dt, 3), "first_run_seconds_incl_setup": round(t_first, 3),
               "steady_frames_per_s": round((args.frames - nq) / max(dt - dq, 1e-9), 2), "setup_seconds": round(dq - nq * (dt - dq) / (args.frames - nq), 3),
               "host_workers_decode_encode": list(host_workers(args.workers)), "streams": args.streams, "png_level": args.png_level,
               "host_stage_ms_per_frame_one_thread": {"decode_resize": round(t_dec * 1e3, 2), "png_encode": round(t_enc * 1e3, 2)},
               "png_megabytes_per_frame": round(png_mb, 2)}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
