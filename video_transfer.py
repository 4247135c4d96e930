#!/usr/bin/env python
"""Video (frame-sequence) style transfer — drop-in for the reference's video_transfer.py (video_transfer.py:17-38,
160-214) on the MI355X HIP path.

Differences that do not change pixels: the style is encoded and factored ONCE (the reference re-encodes it for
every frame, :195); frames move as uint8 through pinned ring buffers and are quantised on the device; decode/resize,
H2D, compute, D2H and encode overlap (vstnet_amd/pipeline.py) instead of running one after the other.  The writer-size quirk of the
reference is kept (:83-86: video_width is overwritten before it scales video_height, so a 1920x1080 clip at
--max_size 1280 is written at 1280x1080 while frames are stylised at 1280x720).
--video may be a directory of frames (always works) or a video file (needs cv2, optional).  Output: an .mp4 if
cv2 is importable, else numbered PNGs.

Multi-GPU (BASELINE config 5; SURVEY 8(e): frames are independent, the host gathers the outputs):
  --gpus N     this process starts N children BEFORE it touches a GPU — child r gets HIP_VISIBLE_DEVICES=r, a CPU thread cap
               of cores // N and `--shard r/N` —, waits for all of them (a failing child fails the run: non-zero exit, the
               others are terminated, nothing is re-executed) and then merges their outputs into the ONE ordered clip / frame
               directory the reference writes (video_transfer.py:92-96,160-214): every frame index present exactly once.
  --shard i/n  process the i-th contiguous shard of the frames (what the children run; also usable by hand).
Every frame is resized on its own like the reference's loop (:161): a frame whose size differs from its predecessor's gets its
own ring buffers and mask plan instead of an error; the writer size comes from the clip's first frame (:82-86).
"""
import sys
import argparse
import os
import re

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

from image_transfer import build_network
from utils.utils import img_resize, load_segment, to_tensor_u8
from vstnet_amd.pipeline import FramePipeline, AsyncSink, prefetch, parallel_map, host_workers, save_png
from vstnet_amd.sharding import shard_range

IMG_EXT = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp')


def build_parser():
    p = argparse.ArgumentParser(allow_abbrev=False)     # (an abbreviated --gpu would survive launch_shards' argv rewrite)
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
    p.add_argument('--gpus', type=int, default=1, help="start this many child processes, one GPU each, and merge their outputs")
    p.add_argument('--precision', type=str, default=None, help="conv arithmetic (default: the library's, bf16x3)")
    p.add_argument('--frames_only', action='store_true', default=False, help="write numbered PNGs even if cv2 is there "
                   "(what the children of --gpus N do: the parent encodes the one clip)")
    p.add_argument('--stub_stylise', action='store_true', default=False, help=argparse.SUPPRESS)   # host-logic tests: no GPU
    p.add_argument('--depth', type=int, default=4, help="pinned ring slots (frames queued ahead of the one being written)")
    p.add_argument('--streams', type=int, default=3, help="frames in flight on the card (1080p, 5-label masks, files in -> PNGs out: "
                   "89 / 102 / 103 frames/s at 2 / 3 / 4, profiles/r04_other_configs.jsonl)")
    p.add_argument('--workers', type=int, default=0, help="host threads that decode + resize input frames, and as many that "
                   "encode output frames (0: a share of this process's cores); the GPU loop itself is one thread")
    p.add_argument('--png_level', type=int, default=0, help="numbered PNGs (the output without cv2, and the shard -> parent "
                   "hand-off of --gpus N): 0 = stored rows, a few ms per 1080p frame, 3 bytes per pixel; 1-9 = PIL's filters + "
                   "zlib at that level (about 30 %% smaller on photographs, 10-30x the encode time).  Lossless either way")
    return p


class FrameDir:
    """A directory of frames as a lazy sequence: a frame is decoded when it is asked for, so a shard (one of N processes) holds
    only its own frames and the parent of a multi-GPU run only the first one (for the writer size)."""

    def __init__(self, path):
        self.files = sorted(os.path.join(path, f) for f in os.listdir(path) if f.lower().endswith(IMG_EXT))

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        return Image.open(self.files[i]).convert('RGB')


def read_frames(path):
    if os.path.isdir(path):
        return FrameDir(path)
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


def clip_name(args):
    return "%s_%s" % (os.path.basename(args.video.rstrip("/")).split(".")[0], os.path.basename(args.style).split(".")[0])


FRAME_PNG = re.compile(r"^\d{5}\.png$")


def launch_shards(args, argv):
    """--gpus N: N children, one per GPU, each on its contiguous shard; decided before this process touches a GPU (no torch.cuda
    call here: GPUs are counted from the visible-devices restriction / the KFD topology, vstnet_amd.sharding.count_gpus)."""
    from vstnet_amd.sharding import launch_children, rank_environment, count_gpus
    n = args.gpus
    base = [a for a in argv]
    for flag in ("--gpus", "--shard"):                  # children get their own --shard and `--gpus 1`
        while flag in base:
            i = base.index(flag)
            del base[i:i + 2]
    base = [a for a in base if not a.startswith("--gpus=") and not a.startswith("--shard=")]
    # (the explicit `--gpus 1` comes last and wins whatever survived the rewrite: a child never launches children)
    cmds = [[sys.executable, os.path.abspath(__file__)] + base + ["--shard", "%d/%d" % (r, n), "--frames_only", "--gpus", "1"]
            for r in range(n)]
    # numbered frames of an earlier (longer) run in the children's output directory would fail the merge only after every child
    # has done its work: they are this script's own outputs, so they go now; anything else in there is left alone
    frame_dir = os.path.join(args.out_dir, clip_name(args))
    if os.path.isdir(frame_dir):
        for f in os.listdir(frame_dir):
            if FRAME_PNG.match(f):
                os.remove(os.path.join(frame_dir, f))
    n_dev = count_gpus()
    return launch_children(cmds, [rank_environment(r, n, visible_device=True, n_devices=n_dev or None) for r in range(n)])


def merge_outputs(frame_dir, n_frames, out_dir, name, fps, size):
    """The host side of SURVEY 8(e): the shards wrote frame i as <frame_dir>/%05d.png; check that every index 0..n-1 is there
    exactly once, then — with cv2 — encode the ONE clip <out_dir>/<name>.mp4 in frame order (one lossy encode, like the
    reference's single writer) and drop the PNGs; without cv2 the ordered frame directory is the output."""
    have = sorted(f for f in os.listdir(frame_dir) if FRAME_PNG.match(f))      # (other files in there are not ours)
    want = ["%05d.png" % i for i in range(n_frames)]
    if have != want:
        missing = sorted(set(want) - set(have))
        extra = sorted(set(have) - set(want))
        raise RuntimeError(f"merge: {len(missing)} frames missing (first {missing[:3]}), {len(extra)} unexpected (first {extra[:3]})")
    try:
        import cv2
    except ImportError:
        return frame_dir
    path = os.path.join(out_dir, name + ".mp4")
    writer = cv2.VideoWriter(path, cv2.VideoWriter_fourcc('m', 'p', '4', 'v'), fps, size)
    try:
        for f in want:
            writer.write(np.asarray(Image.open(os.path.join(frame_dir, f)).convert('RGB'))[..., ::-1])
    finally:
        writer.release()
    for f in want:
        os.remove(os.path.join(frame_dir, f))
    if not os.listdir(frame_dir):
        os.rmdir(frame_dir)
    return path


class _SizeContext:
    """Everything that depends on the stylised frame size: ring buffers / streams (FramePipeline), the mask plan, the decode
    hook that resizes to the writer size.  One per distinct size met in the clip (normally exactly one)."""

    def __init__(self, args, net, cwct, z_s, s_stats, style_seg, size_wh, writer_wh, device):
        cw_, ch_ = size_wh
        video_width, video_height = writer_wh
        masked = style_seg is not None
        self.size = size_wh
        plan = None
        if masked:      # one label map for every frame and one style: histograms, uploads and the style side happen once
            content_seg = load_segment(args.content_seg, size_wh)[None, ...]
            with torch.no_grad():
                zc_shape = (1, 32, ch_, cw_) if net.sp_steps == 2 else (1, 128, ch_ // 2, cw_ // 2)
                # (learn_slots: one read-back per size; with at most 8 labels the masked transfer then stays on the packed code)
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
        self.pipe = FramePipeline(net, transform, ch_, cw_, device=device, depth=args.depth, compute_streams=args.streams,
                                  decode=decode, out_height=video_height, out_width=video_width)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    if args.auto_seg:
        raise NotImplementedError("--auto_seg needs mmseg/SegFormer (not part of this repository)")
    os.makedirs(args.out_dir, exist_ok=True)
    name = clip_name(args)
    frames = read_frames(args.video)
    video_width, video_height = writer_size(frames[0], args.max_size)

    if args.gpus > 1:                                   # parent of a multi-GPU run: never initialises a GPU itself
        rc = launch_shards(args, argv)
        if rc != 0:
            raise SystemExit(rc)
        out = merge_outputs(os.path.join(args.out_dir, name), len(frames), args.out_dir, name, args.fps, (video_width, video_height))
        print("Save stylized video at %s" % out)
        return out

    rank, world = (int(v) for v in args.shard.split("/"))
    lo, hi = shard_range(len(frames), rank, world)
    down_scale = 4
    net = cwct = z_s = s_stats = style_seg = device = None
    masked = args.content_seg is not None and args.style_seg is not None
    if not args.stub_stylise:
        device = torch.device("cuda")
        net = build_network(args.mode, args.ckpoint, args.synthetic_weights, device, args.precision)
        down_scale = net.down_scale
        from models.cWCT import cWCT
        cwct = cWCT(precision=args.precision)
        style = img_resize(Image.open(args.style).convert('RGB'), args.max_size, down_scale=down_scale)
        with torch.no_grad():
            z_s = net.forward_u8(to_tensor_u8(style).to(device))
            s_stats = cwct.style_stats(z_s) if not masked and args.alpha_c is None else None
        style_seg = load_segment(args.style_seg, style.size)[None, ...] if masked else None

    writer, frame_dir = None, None
    cv2 = None
    if not args.frames_only:
        try:
            import cv2
        except ImportError:
            cv2 = None
    if cv2 is not None:
        writer = cv2.VideoWriter(os.path.join(args.out_dir, name + (".mp4" if world == 1 else "_%d.mp4" % rank)),
                                 cv2.VideoWriter_fourcc('m', 'p', '4', 'v'), args.fps, (video_width, video_height))
    else:
        frame_dir = os.path.join(args.out_dir, name)
        os.makedirs(frame_dir, exist_ok=True)

    dec_workers, enc_workers = host_workers(args.workers)

    def write(i, out):
        if writer is not None:
            writer.write(out[..., ::-1])
        else:
            save_png(os.path.join(frame_dir, "%05d.png" % i), out, args.png_level)

    def load(i):         # decode + resize; EVERY frame is resized on its own (video_transfer.py:161)
        return i, np.asarray(img_resize(frames[i], args.max_size, down_scale=down_scale), dtype=np.uint8)

    def source():        # background threads, frames in order
        return parallel_map(load, range(lo, hi), workers=dec_workers if isinstance(frames, FrameDir) else 1, ahead=args.depth)

    # numbered PNGs are independent files: encode them on several threads; a video writer takes its frames in order from one
    sink = AsyncSink(write, ahead=2 * enc_workers, workers=1 if writer is not None else enc_workers)
    try:
        if args.stub_stylise:       # host-logic rehearsal: the "stylised" frame is the resized frame at the writer size
            for i, arr in source():
                sink(i, np.asarray(Image.fromarray(arr).resize((video_width, video_height), Image.BICUBIC)))
        else:
            # consecutive frames of one size stream through that size's pipeline; a size change (rare: the reference resizes
            # every frame on its own, :161) drains it and switches to the other size's context
            contexts = {}
            it = iter(prefetch(source(), ahead=args.depth))
            pending = next(it, None)
            while pending is not None:
                size_wh, start = (pending[1].shape[1], pending[1].shape[0]), pending[0]

                def same_size_run():
                    nonlocal pending
                    while pending is not None and (pending[1].shape[1], pending[1].shape[0]) == size_wh:
                        arr = pending[1]
                        pending = next(it, None)
                        yield arr
                ctx = contexts.get(size_wh)
                if ctx is None:
                    ctx = contexts[size_wh] = _SizeContext(args, net, cwct, z_s, s_stats, style_seg, size_wh,
                                                           (video_width, video_height), device)
                ctx.pipe.run(same_size_run(), sink, start_index=start)
    finally:
        try:
            sink.close()
        finally:
            if writer is not None:
                writer.release()
                writer = None
    print("Save stylized video at %s" % (frame_dir or args.out_dir))
    return frame_dir or args.out_dir


if __name__ == "__main__":
    main()
