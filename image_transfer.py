#!/usr/bin/env python
"""Single-image style transfer — drop-in for the reference's image_transfer.py (same flags and call sequence,
image_transfer.py:15-37,172-221) on the MI355X HIP path; no torchvision / todos / pdb.

    python image_transfer.py --mode photorealistic --ckpoint checkpoints/photo_image.pt \
        --content data/content/01.jpg --style data/style/01.jpg [--alpha_c 0.3] [--content_seg c.png --style_seg s.png]

--auto_seg needs the external SegFormer stack (mmseg + weights), which is outside this repository's scope.
--synthetic_weights runs with the deterministic synthetic checkpoint (no trained checkpoint ships with the repo).
"""
import argparse
import os

import numpy as np
import torch
from PIL import Image

from utils.utils import img_resize, load_segment, to_tensor_u8


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--mode', type=str, default='photorealistic')
    p.add_argument('--ckpoint', type=str, default='checkpoints/photo_image.pt')
    p.add_argument('--content', type=str, default='data/content/01.jpg')
    p.add_argument('--style', type=str, default='data/style/01.jpg')
    p.add_argument('--out_dir', type=str, default="output")
    p.add_argument('--max_size', type=int, default=1280)
    p.add_argument('--alpha_c', type=float, default=None)
    p.add_argument('--content_seg', type=str, default=None)
    p.add_argument('--style_seg', type=str, default=None)
    p.add_argument('--auto_seg', action='store_true', default=False)
    p.add_argument('--synthetic_weights', action='store_true', default=False)
    p.add_argument('--precision', type=str, default=None, help="conv arithmetic (default: the library's, bf16x3)")
    # the delldu fork's post-process (project/image_style/vstnet.py:189-220): keep the content's Lab luminance
    p.add_argument('--preserve_luminance', action='store_true', default=False)
    return p


def build_network(mode, ckpoint, synthetic, device, precision=None):
    from models.RevResNet import RevResNet
    if mode.lower() == "photorealistic":
        hd, sp = 16, 2
    elif mode.lower() == "artistic":
        hd, sp = 64, 1
    else:
        raise NotImplementedError()
    net = RevResNet(hidden_dim=hd, sp_steps=sp, precision=precision)
    if synthetic:
        from vstnet_amd.synth import synthetic_state_dict
        net.load_state_dict(synthetic_state_dict(1234, hd, sp))
    else:
        state_dict = torch.load(ckpoint, map_location="cpu", weights_only=True)
        net.load_state_dict(state_dict['state_dict'])
    return net.to(device).eval()


def stylize(net, cwct, content_img, style_img, content_seg=None, style_seg=None, alpha_c=None, device="cuda",
            preserve_luminance=False):
    """image_transfer.py:172-201 with the uint8 frame edge on the device; returns uint8 [H,W,3] numpy."""
    with torch.no_grad():
        z_c = net.forward_u8(to_tensor_u8(content_img).to(device))
        z_s = net.forward_u8(to_tensor_u8(style_img).to(device))
        if alpha_c is not None and content_seg is None and style_seg is None:
            assert 0.0 <= alpha_c <= 1.0
            z_cs = cwct.interpolation(z_c, styl_feat_list=[z_s], alpha_s_list=[1.0], alpha_c=alpha_c)
        else:
            z_cs = cwct.transfer(z_c, z_s, content_seg, style_seg)
        if not preserve_luminance:
            return net.inverse_u8(z_cs)[0].cpu().numpy()
        from vstnet_amd.color import luminance_transfer
        content = to_tensor_u8(content_img).to(device).permute(0, 3, 1, 2).float().div(255.0)
        out = luminance_transfer(content, net(z_cs, forward=False))
        return out[0].mul(255.0).clamp(0, 255).byte().permute(1, 2, 0).cpu().numpy()


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.auto_seg:
        raise NotImplementedError("--auto_seg needs mmseg/SegFormer (not part of this repository); pass --content_seg/--style_seg")
    device = torch.device("cuda")
    os.makedirs(args.out_dir, exist_ok=True)
    net = build_network(args.mode, args.ckpoint, args.synthetic_weights, device, args.precision)
    from models.cWCT import cWCT
    cwct = cWCT(precision=args.precision)

    content = Image.open(args.content).convert('RGB')
    style = Image.open(args.style).convert('RGB')
    content = img_resize(content, args.max_size, down_scale=net.down_scale)
    style = img_resize(style, args.max_size, down_scale=net.down_scale)
    content_seg = style_seg = None
    if args.content_seg is not None and args.style_seg is not None:
        content_seg = load_segment(args.content_seg, content.size)[None, ...]
        style_seg = load_segment(args.style_seg, style.size)[None, ...]

    out = stylize(net, cwct, content, style, content_seg, style_seg, args.alpha_c, device, args.preserve_luminance)
    cn, sn = os.path.basename(args.content), os.path.basename(args.style)
    path = os.path.join(args.out_dir, "%s_%s.png" % (cn.split(".")[0], sn.split(".")[0]))
    Image.fromarray(out).save(path, quality=100)
    print("Save at %s" % path)
    return path


if __name__ == "__main__":
    main()
