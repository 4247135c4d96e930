#!/usr/bin/env python
"""Video (frame-sequence) style transfer — drop-in for the reference's video_transfer.py (video_transfer.py:17-38,
160-214) on the MI355X HIP path.

Differences that do not change pixels: the style is encoded and factored ONCE (the reference re-encodes it for
every frame, :195); frames move as uint8 through pinned ring buffers and are quantised on the device; decode/resize,
H2D, compute, D2H and encode overlap (vstnet_amd/pipeline.py) instead of running one after the other.  The writer-size quirk of the
reference is kept (:83-86: video_width is overwritten before it scales video_height, so a 1920x1080 clip at
--max_size 1280 is written at 1280x1080 while frames are stylised at 1280x720).
--video may be a directory of frames (always works) or a video file (needs cv2, optional).  Output: an .mp4 if
cv2 is importable, else numbered PNGs.  --shard i/n processes the i-th contiguous shard of the frames (one
process per GPU; frames are independent).
"""
import argparse
import os

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

from image_transfer import build_network
from utils.utils import img_resize, load_segment, to_tensor_u8
from vstnet_amd.pipeline import FramePipeline, AsyncSink, prefetch
from vstnet_amd.sharding import shard_range

IMG_EXT = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp')


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--mode', type=str, default='photorealistic')
    p.add_argument('--ckpoint', type=str, default='checkpoints/photo_video.pt')
    p.add_argument('--video', type=str, default='data/content/03.avi')
    p.add_argument('--style', type=str, default='data/style/03.jpeg')
    p.add_argument('--out_dir', type=str, default="output")
    p.add_argument('--max_size', type=int, default=1280)
    p.add_argument('--alpha_c', type=float, default=None)
    p.add_argument('--fps', type=int, default=30)
    p.add_argument('--content_seg', type=str, default=None, help="one label map used for every frame")
    p.add_argument('--style_seg', type=str, default=None)
    p.add_argument('--auto_seg', action='store_true', default=False)
    p.add_argument('--synthetic_weights', action='store_true', default=False)
    p.add_argument('--shard', type=str, default="0/1")
    p.add_argument('--depth', type=int, default=4, help="pinned ring slots (frames queued ahead of the one being written)")
    p.add_argument('--streams', type=int, default=2, help="frames in flight on the card")
    return p


def read_frames(path):
    if os.path.isdir(path):
        files = sorted(os.path.join(path, f) for f in os.listdir(path) if f.lower().endswith(IMG_EXT))
        return [Image.open(f).convert('RGB') for f in files]
    try:
        import cv2
    except ImportError as e:
        raise RuntimeError("reading a video file needs cv2; pass a directory of frames instead") from e
    frames, cap = [], cv2.VideoCapture(path)
    while True:
        ret, frame = cap.read()
        if ret is False:
            break
        frames.append(Image.fromarray(frame[..., ::-1]))
    return frames


def writer_size(first_frame, max_size):
    """video_transfer.py:82-86, including its overwrite-before-use quirk."""
    video_height, video_width = np.array(first_frame).shape[:2]
    if max(video_width, video_height) > max_size:
        video_width = int(1.0 * video_width / max(video_width, video_height) * max_size)
        video_height = int(1.0 * video_height / max(video_width, video_height) * max_size)
    return video_width, video_height


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.auto_seg:
        raise NotImplementedError("--auto_seg needs mmseg/SegFormer (not part of this repository)")
    device = torch.device("cuda")
    os.makedirs(args.out_dir, exist_ok=True)
    net = build_network(args.mode, args.ckpoint, args.synthetic_weights, device)
    from models.cWCT import cWCT
    cwct = cWCT()

    frames = read_frames(args.video)
    rank, world = (int(v) for v in args.shard.split("/"))
    lo, hi = shard_range(len(frames), rank, world)
    video_width, video_height = writer_size(frames[0], args.max_size)

    style = img_resize(Image.open(args.style).convert('RGB'), args.max_size, down_scale=net.down_scale)
    masked = args.content_seg is not None and args.style_seg is not None
    with torch.no_grad():
        z_s = net.forward_u8(to_tensor_u8(style).to(device))
        s_stats = cwct.style_stats(z_s) if not masked and args.alpha_c is None else None
    style_seg = load_segment(args.style_seg, style.size)[None, ...] if masked else None

    name = "%s_%s" % (os.path.basename(args.video.rstrip("/")).split(".")[0], os.path.basename(args.style).split(".")[0])
    writer, frame_dir = None, None
    try:
        import cv2
        writer = cv2.VideoWriter(os.path.join(args.out_dir, name + (".mp4" if world == 1 else "_%d.mp4" % rank)),
                                 cv2.VideoWriter_fourcc('m', 'p', '4', 'v'), args.fps, (video_width, video_height))
    except ImportError:
        frame_dir = os.path.join(args.out_dir, name)
        os.makedirs(frame_dir, exist_ok=True)

    def write(i, out):
        if writer is not None:
            writer.write(out[..., ::-1])
        else:
            Image.fromarray(out).save(os.path.join(frame_dir, "%05d.png" % i))

    if hi > lo:
        first = img_resize(frames[lo], args.max_size, down_scale=net.down_scale)
        content_seg = load_segment(args.content_seg, first.size)[None, ...] if masked else None
        cw_, ch_ = first.size
        plan = None
        if masked:      # one label map for every frame and one style: histograms, uploads and the style side happen once
            with torch.no_grad():
                zc_shape = (1, 32, ch_, cw_) if net.sp_steps == 2 else (1, 128, ch_ // 2, cw_ // 2)
                # (learn_slots: one read-back per clip; with at most 8 labels the masked transfer then stays on the packed code)
                plan = cwct.bind_style(cwct.learn_slots(cwct.plan_masks(content_seg, style_seg, zc_shape, z_s.shape, device)), z_s)

        def transform(z_c, i):
            if args.alpha_c is not None and not masked:
                assert 0.0 <= args.alpha_c <= 1.0
                return cwct.interpolation(z_c, styl_feat_list=[z_s], alpha_s_list=[1.0], alpha_c=args.alpha_c)
            if masked:
                return cwct.transfer_with_plan(z_c, None, plan)
            return cwct.transfer_with_stats(z_c, s_stats)

        decode = None
        if (cw_, ch_) != (video_width, video_height):
            def decode(z_cs):   # transforms.Resize((video_height, video_width), BICUBIC) on the float tensor, then quantise
                sty = net(z_cs, forward=False)
                sty = F.interpolate(sty, size=(video_height, video_width), mode="bicubic", align_corners=False, antialias=True)
                return sty.mul(255).clamp(0, 255).byte().permute(0, 2, 3, 1).contiguous()

        def source():        # decode + resize in a background thread (img_resize: utils/utils.py:90-101)
            for i in range(lo, hi):
                arr = np.asarray(img_resize(frames[i], args.max_size, down_scale=net.down_scale), dtype=np.uint8)
                if arr.shape[:2] != (ch_, cw_):     # the pinned ring, the mask plan and the writer are sized once per clip
                    raise ValueError(f"frame {i} resizes to {arr.shape[1]}x{arr.shape[0]}, the clip's first frame to "
                                     f"{cw_}x{ch_}: frames of one clip must share a size")
                yield arr

        pipe = FramePipeline(net, transform, ch_, cw_, device=device, depth=args.depth, compute_streams=args.streams,
                             decode=decode, out_height=video_height, out_width=video_width)
        sink = AsyncSink(write)
        try:
            pipe.run(prefetch(source(), ahead=args.depth), sink, start_index=lo)
        finally:
            try:
                sink.close()
            finally:
                if writer is not None:
                    writer.release()
                    writer = None
    if writer is not None:
        writer.release()
    print("Save stylized video at %s" % (frame_dir or args.out_dir))
    return frame_dir or args.out_dir


if __name__ == "__main__":
    main()
