"""ORACLE — test infrastructure, NOT product code.

A from-scratch CPU (torch fp32 / fp64, ATen ops only) restatement of the
reference's inference hot path: the RevResNet reversible encoder/decoder and the
cWCT Cholesky whitening/colouring transform.  It is functional (no nn.Module):
every function takes the plain ``state_dict`` of the reference
(``stack.{i}.conv.{1,4,7}.{weight,bias}``, ...).

Who may import this file: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — as the checker / the timed CPU baseline.
The product path (``vstnet_amd``) never imports it and has no CPU fallback.

Pinning: ``oracle/make_golden.py`` imports the real reference modules
(/root/reference/models/{RevResNet,cWCT}.py, two shims) in
the build container, checks this restatement against them and writes
``tests/golden/*.npz``.  ``tests/test_oracle.py`` re-checks this file against
those fixtures wherever it runs (the reference itself never travels).

Every function cites the reference lines it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

STACK = [(1, 16)] * 10 + [(2, 64)] + [(1, 64)] * 9 + [(2, 256)] + [(1, 256)] * 9


# --------------------------------------------------------------------------- R-1..R-3 glue
def split(x):
    """models/RevResNet.py:8-12 — channel halves."""
    n = x.shape[1] // 2
    return x[:, :n].contiguous(), x[:, n:].contiguous()


def merge(x1, x2):
    """models/RevResNet.py:15-16."""
    return torch.cat((x1, x2), dim=1)


def inj_pad_fwd(x, pad):
    """models/RevResNet.py:25-28 — append `pad` zero channels."""
    if pad == 0:
        return x
    z = x.new_zeros(x.shape[0], pad, x.shape[2], x.shape[3])
    return torch.cat((x, z), dim=1)


def inj_pad_inv(x, pad):
    """models/RevResNet.py:30-31 — drop the last `pad` channels."""
    return x[:, : x.shape[1] - pad]


def squeeze(x):
    """models/RevResNet.py:34-37 — out[b,(i*2+j)*D+d,h,w] = in[b,d,2h+i,2w+j]."""
    b, d, h, w = x.shape
    if h % 2 or w % 2:
        raise RuntimeError("squeeze needs even H and W")
    parts = [x[:, :, i::2, j::2] for i in (0, 1) for j in (0, 1)]
    return torch.cat(parts, dim=1).contiguous()


def unsqueeze(x):
    """models/RevResNet.py:40-43 — exact inverse of squeeze."""
    b, c, h, w = x.shape
    d = c // 4
    out = x.new_empty(b, d, 2 * h, 2 * w)
    for i in (0, 1):
        for j in (0, 1):
            k = i * 2 + j
            out[:, :, i::2, j::2] = x[:, k * d:(k + 1) * d]
    return out


# --------------------------------------------------------------------------- R-4 / R-5 block
def _conv(x, w, b, stride=1):
    # ReflectionPad2d((k-1)//2) + Conv2d(k, padding=0, bias=True): models/RevResNet.py:73,79-88 (k = 3 in the published nets)
    p = (w.shape[-1] - 1) // 2
    return F.conv2d(F.pad(x, (p, p, p, p), mode="reflect") if p else x, w, b, stride=stride)


def residual_F(x2, sd, prefix, stride):
    """F(x2) of models/RevResNet.py:79-88 (three reflect-padded 3x3 convs, ReLU between)."""
    h = F.relu(_conv(x2, sd[prefix + "conv.1.weight"], sd[prefix + "conv.1.bias"], stride))
    h = F.relu(_conv(h, sd[prefix + "conv.4.weight"], sd[prefix + "conv.4.bias"]))
    return _conv(h, sd[prefix + "conv.7.weight"], sd[prefix + "conv.7.bias"])


def block_forward(x1, x2, sd, prefix, stride):
    """models/RevResNet.py:96-104 — (x1,x2) -> (x2', F(x2)+x1')."""
    fx2 = residual_F(x2, sd, prefix, stride)
    if stride == 2:
        x1, x2 = squeeze(x1), squeeze(x2)
    return x2, fx2 + x1


def block_inverse(x2, y1, sd, prefix, stride):
    """models/RevResNet.py:106-116 — (x2', y1) -> (x1, x2)."""
    if stride == 2:
        x2 = unsqueeze(x2)
    x1 = y1 - residual_F(x2, sd, prefix, stride)
    if stride == 2:
        x1 = unsqueeze(x1)
    return x1, x2


# --------------------------------------------------------------------------- R-6 / R-7 network
def arch_stack(arch):
    """[(stride, channel)] of the stack for a constructor-argument set (block_stack, models/RevResNet.py:190-201);
    arch = dict(nBlocks, nStrides, nChannels, ...), None = the published architecture."""
    if arch is None:
        return STACK
    out = []
    for ch, depth, st in zip(arch["nChannels"], arch["nBlocks"], arch["nStrides"]):
        out += [(st, ch)] + [(1, ch)] * (depth - 1)
    return out


def revnet_forward(x, sd, sp_steps=2, arch=None):
    """models/RevResNet.py:210-223 + channel_reduction.forward :131-146.  arch (optional): the reference's other constructor
    arguments — dict(nBlocks, nStrides, nChannels, hidden_dim[, in_channel]); mult and kernel are implied by the weight shapes."""
    stack = arch_stack(arch)
    c0 = stack[0][1]
    x = inj_pad_fwd(x, 2 * c0 - x.shape[1])
    x1, x2 = split(x)
    for i, (stride, _) in enumerate(stack):
        x1, x2 = block_forward(x1, x2, sd, f"stack.{i}.", stride)
    x1, x2 = split(merge(x1, x2))               # channel_reduction.forward: split, inj_pad (0 in the published nets)
    cr_pad = 0 if arch is None else arch["hidden_dim"] * 4 ** sp_steps - stack[-1][1]
    x1, x2 = inj_pad_fwd(x1, cr_pad), inj_pad_fwd(x2, cr_pad)
    for i in range(2):
        x1, x2 = block_forward(x1, x2, sd, f"channel_reduction.block_list.{i}.", 1)
    z = merge(x1, x2)
    for _ in range(sp_steps):                   # "spread" :141-144 == unsqueeze
        z = unsqueeze(z)
    return z


def revnet_inverse(z, sd, sp_steps=2, in_channel=3, arch=None):
    """models/RevResNet.py:225-239 + channel_reduction.inverse :148-163."""
    stack = arch_stack(arch)
    for _ in range(sp_steps):
        z = squeeze(z)
    a, b = split(z)
    for i in (1, 0):
        a, b = block_inverse(a, b, sd, f"channel_reduction.block_list.{i}.", 1)
    cr_pad = 0 if arch is None else arch["hidden_dim"] * 4 ** sp_steps - stack[-1][1]
    a, b = inj_pad_inv(a, cr_pad), inj_pad_inv(b, cr_pad)
    a, b = split(merge(a, b))
    for i in range(len(stack) - 1, -1, -1):
        a, b = block_inverse(a, b, sd, f"stack.{i}.", stack[i][0])
    return inj_pad_inv(merge(a, b), 2 * stack[0][1] - in_channel)


# --------------------------------------------------------------------------- C-4 Cholesky
def cholesky_dec(conv, eps=2e-5, invert=False, return_tries=False):
    """models/cWCT.py:111-132 — Cholesky with the cumulative jitter schedule
    (after k failed retries the total added jitter is eps*k(k+1)/2), optional general inverse."""
    tries = 0
    L, info = torch.linalg.cholesky_ex(conv)
    if int(info.max()) != 0:
        iden = torch.eye(conv.shape[-1])        # float32 like the reference's (models/cWCT.py:120), also under use_double:
                                                # the float64 covariance then receives eps rounded to float32
        e = eps
        while True:
            conv = conv + iden * e
            tries += 1
            L, info = torch.linalg.cholesky_ex(conv)
            if int(info.max()) == 0:
                break
            e = e + eps
    if invert:
        L = torch.inverse(L)
    return (L, tries) if return_tries else L


# --------------------------------------------------------------------------- C-2 / C-3
def whitening(x, eps=2e-5):
    """models/cWCT.py:134-149 on a 2-D [N,L] matrix."""
    xc = x - x.mean(-1, keepdim=True)
    conv = (xc @ xc.transpose(-1, -2)) / (x.shape[-1] - 1)
    return cholesky_dec(conv, eps, invert=True) @ xc


def coloring(whiten, style, eps=2e-5):
    """models/cWCT.py:152-164."""
    mu = style.mean(-1, keepdim=True)
    sc = style - mu
    conv = (sc @ sc.transpose(-1, -2)) / (style.shape[-1] - 1)
    return cholesky_dec(conv, eps, invert=False) @ whiten + mu


def transfer(content, style, eps=2e-5, use_double=False):
    """C-1: the INTENDED no-mask semantics of models/cWCT.py:24-47 (the fork's batched `whitening` raises on 3-D input,
    SURVEY.md 8(a) C-1): coloring(whitening(c), s) on the whole [B,N,L] batch, which is what `interpolation(c, [s],
    [1.0], 0.0)` computes (models/cWCT.py:206-262) — including its batch-wide Cholesky jitter (see `interpolation`)."""
    return interpolation(content, [style], [1.0], 0.0, eps, use_double)


# --------------------------------------------------------------------------- C-5 masked
def compute_label_info(cseg, sseg):
    """models/cWCT.py:166-189 — labels of the content mask and their validity
    (count_c>10, count_s>10, ratio<100 both ways)."""
    cseg = np.asarray(cseg)
    sseg = np.asarray(sseg)
    label_set = np.unique(cseg)
    indicator = np.zeros(int(cseg.max()) + 1)
    for l in label_set:
        a = int((cseg == l).sum())
        b = int((sseg == l).sum())
        indicator[l] = a > 10 and b > 10 and a / b < 100 and b / a < 100
    return label_set, indicator


def transfer_seg(content, style, cmask, smask, eps=2e-5, use_double=False):
    """models/cWCT.py:49-109 — per label: gather columns, whiten, colour, scatter back.
    Masks are used at feature resolution (not resized in this fork, :72-73)."""
    B, N, H, W = content.shape
    dt = content.dtype
    c = content.reshape(B, N, -1).clone()
    s = style.reshape(B, N, -1)
    if use_double:
        c, s = c.double(), s.double()
    for b in range(B):
        labels, ok = compute_label_info(cmask[b], smask[b])
        cm = torch.from_numpy(np.asarray(cmask[b]).reshape(-1).astype(np.int64))
        sm = torch.from_numpy(np.asarray(smask[b]).reshape(-1).astype(np.int64))
        src = c[b]
        tgt = src.clone()
        for l in labels:
            if not ok[l]:
                continue
            ci = torch.nonzero(cm == int(l)).reshape(-1)
            si = torch.nonzero(sm == int(l)).reshape(-1)
            if ci.numel() == 0 or si.numel() == 0:
                continue
            tgt[:, ci] = coloring(whitening(src[:, ci], eps), s[b][:, si], eps)
        c[b] = tgt
    return c.to(dt).reshape(B, N, H, W)


# --------------------------------------------------------------------------- C-6 interpolation
def interpolation(content, styles, alphas, alpha_c=0.0, eps=2e-5, use_double=False):
    """models/cWCT.py:206-262 — whiten content once, mix the styles' Cholesky factors and means, optionally blend with
    the content's own factor/mean.  Everything is batched over B like the reference, so `cholesky_dec` sees the [B,N,N]
    stack: if ANY sample's covariance fails, the jitter is added to EVERY sample's (models/cWCT.py:115-128)."""
    assert len(styles) == len(alphas)
    B, N, H, W = content.shape
    dt = content.dtype
    c = content.reshape(B, N, -1)
    if use_double:
        c = c.double()
    cmean = c.mean(-1)
    cc = c - cmean.unsqueeze(-1)
    conv = (cc @ cc.transpose(-1, -2)) / (cc.shape[-1] - 1)
    whiten = cholesky_dec(conv, eps, invert=True) @ cc
    mixL = torch.zeros_like(conv)
    mixm = torch.zeros_like(cmean)
    for sf, a in zip(styles, alphas):
        assert sf.shape[0] == B and sf.shape[1] == N
        s = sf.reshape(B, N, -1)
        if use_double:
            s = s.double()
        sm = s.mean(-1)
        sc = s - sm.unsqueeze(-1)
        sconv = (sc @ sc.transpose(-1, -2)) / (s.shape[-1] - 1)
        mixL = mixL + cholesky_dec(sconv, eps) * a
        mixm = mixm + sm * a
    if alpha_c != 0.0:
        mixL = mixL * (1 - alpha_c) + cholesky_dec(conv, eps) * alpha_c
        mixm = mixm * (1 - alpha_c) + cmean * alpha_c
    out = mixL @ whiten + mixm.unsqueeze(-1)
    return out.to(dt).reshape(B, N, H, W)


# --------------------------------------------------------------------------- whole stylisation (H-1)
def stylize(content, style, sd, sp_steps=2, cmask=None, smask=None, alpha_c=None):
    """image_transfer.py:172-201 — fwd, fwd, transfer|interpolation, inv."""
    zc = revnet_forward(content, sd, sp_steps)
    zs = revnet_forward(style, sd, sp_steps)
    if alpha_c is not None and cmask is None and smask is None:
        zcs = interpolation(zc, [zs], [1.0], alpha_c)
    elif cmask is None or smask is None:
        zcs = transfer(zc, zs)
    else:
        zcs = transfer_seg(zc, zs, cmask, smask)
    return zc, zs, zcs, revnet_inverse(zcs, sd, sp_steps)


def to_uint8(img):
    """image_transfer.py:217-218 — mul(255).clamp(0,255).byte() (truncation), NCHW -> NHWC."""
    return img.mul(255).clamp(0, 255).byte().permute(0, 2, 3, 1).contiguous()


# --------------------------------------------------------------------------- host helpers of the scripts (H-1 edge)
def colors_to_labels_loop(arr):
    """utils/utils.py:105-136 restated as the reference's per-pixel loop (small images only): exact colour match,
    else the nearest colour in L1; on a tie the reference's branch raises inside try/except, keeping the first."""
    color_dict = {(0, 0, 255): 3, (0, 255, 0): 2, (0, 0, 0): 0, (255, 255, 255): 1, (255, 0, 0): 4, (255, 255, 0): 5,
                  (128, 128, 128): 6, (0, 255, 255): 7, (255, 0, 255): 8}
    arr = np.asarray(arr)
    out = np.zeros(arr.shape[:-1])
    for x in range(arr.shape[0]):
        for y in range(arr.shape[1]):
            px = tuple(int(v) for v in arr[x, y, :])
            if px in color_dict:
                out[x, y] = color_dict[px]
                continue
            best, best_d = 0, 99999
            for key, val in color_dict.items():
                d = int(np.sum(np.abs(np.asarray(key) - arr[x, y, :].astype(np.int64))))
                if d < best_d:
                    best_d, best = d, val
            out[x, y] = best
    return out.astype(np.uint8)


class SegReMappingLoop:
    """models/segmentation/SegReMapping.py:5-76 restated with the reference's per-label `seg == label` scans."""

    def __init__(self, label_mapping, min_ratio=0.01):
        self.label_mapping = np.asarray(label_mapping)
        self.min_ratio = min_ratio

    def cross_remapping(self, content_seg, style_seg):
        cont = list(np.unique(content_seg))
        style = list(np.unique(style_seg))
        new = list(cont)
        for s in set(cont) - set(style):
            for j in range(self.label_mapping.shape[0]):
                cand = self.label_mapping[j, s]
                if cand in style:
                    new[cont.index(s)] = cand
                    break
        out = content_seg.copy()
        for i, cur in enumerate(cont):
            out[content_seg == cur] = new[i]
        return out

    def self_remapping(self, seg):
        out = seg.copy()
        n = seg.shape[0] * seg.shape[1]
        labels, ratios = [], []
        for l in np.unique(seg):
            labels.append(l)
            ratios.append(np.sum(np.float32(seg == l)) / n)
        new = list(labels)
        for i, cur in enumerate(labels):
            if ratios[i] < self.min_ratio:
                for j in range(self.label_mapping.shape[0]):
                    cand = self.label_mapping[j, cur]
                    if cand in labels and ratios[labels.index(cand)] >= self.min_ratio:
                        new[i] = cand
                        break
        for i, cur in enumerate(labels):
            out[seg == cur] = new[i]
        return out


# --------------------------------------------------------------------------- Lab luminance post-process (8(f) rank 4)
def _rgb2lab(rgb):
    """project/image_style/color.py:18-53,94-104 — sRGB [0,1] -> Lab rescaled to [-1,1] (clamped).  Same arithmetic
    order as the reference (mask-multiply selects, left-to-right sums)."""
    mask = (rgb > 0.04045).float()
    rgb = (((rgb + 0.055) / 1.055) ** 2.4) * mask + rgb / 12.92 * (1.0 - mask)
    r, g, b = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    x = 0.412453 * r + 0.357580 * g + 0.180423 * b
    y = 0.212671 * r + 0.715160 * g + 0.072169 * b
    z = 0.019334 * r + 0.119193 * g + 0.950227 * b
    xyz = torch.stack((x, y, z), dim=1)
    sc = torch.tensor((0.95047, 1.0, 1.08883))[None, :, None, None]
    t = xyz / sc
    mask = (t > 0.008856).float()
    f = t ** (1.0 / 3.0) * mask + (7.787 * t + 16.0 / 116.0) * (1.0 - mask)
    L = 116.0 * f[:, 1] - 16.0
    a = 500.0 * (f[:, 0] - f[:, 1])
    b = 200.0 * (f[:, 1] - f[:, 2])
    lab = torch.stack((L, a, b), dim=1)
    out = torch.cat(((lab[:, 0:1] - 50.0) / 50.0, lab[:, 1:3] / 110.0), dim=1)
    return out.clamp(-1.0, 1.0)


def _lab2rgb(lab_rs):
    """project/image_style/color.py:56-91,107-113."""
    L = lab_rs[:, 0] * 50.0 + 50.0
    a = lab_rs[:, 1] * 110.0
    b = lab_rs[:, 2] * 110.0
    y = (L + 16.0) / 116.0
    x = (a / 500.0) + y
    z = torch.max(torch.tensor((0,)), y - (b / 200.0))
    out = torch.stack((x, y, z), dim=1)
    mask = (out > 0.2068966).float()
    out = (out ** 3.0) * mask + (out - 16.0 / 116.0) / 7.787 * (1.0 - mask)
    xyz = out * torch.tensor((0.95047, 1.0, 1.08883))[None, :, None, None]
    X, Y, Z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    r = 3.24048134 * X - 1.53715152 * Y - 0.49853633 * Z
    g = -0.96925495 * X + 1.87599 * Y + 0.04155593 * Z
    bl = 0.05564664 * X - 0.20404134 * Y + 1.05731107 * Z
    rgb = torch.stack((r, g, bl), dim=1)
    rgb = torch.max(rgb, torch.zeros_like(rgb))
    mask = (rgb > 0.0031308).float()
    rgb = (1.055 * (rgb ** (1.0 / 2.4)) - 0.055) * mask + 12.92 * rgb * (1.0 - mask)
    return rgb.clamp(0.0, 1.0)


def luminance_transfer(content, stylized):
    """project/image_style/vstnet.py:189-220 — keep the content's L channel, take a/b from the (clamped) stylised
    image: lab2rgb(cat(L(content), ab(stylized)))."""
    lab_c = _rgb2lab(content)
    lab_o = _rgb2lab(stylized.clamp(0.0, 1.0))        # the fork's decoder clamps its output (vstnet.py:322)
    return _lab2rgb(torch.cat((lab_c[:, 0:1], lab_o[:, 1:3]), dim=1))
