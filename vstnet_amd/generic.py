"""RevResNet for architectures other than the published one (models/RevResNet.py:166-239 with arbitrary nBlocks / nStrides /
nChannels / mult / kernel / in_channel / hidden_dim / sp_steps): the reference's own sequence of ops, each one a HIP kernel of
csrc/generic.hip on plain NCHW fp32 tensors (exact fp32 FMA).  Complete, not fast: the tuned kernels exist for the published
architecture only (vstnet_amd/revresnet.py picks the path).  torch allocates the tensors; no torch op computes anything here."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def conv(x, weight, bias, stride=1, relu=False, old=None, sign=1.0, out=None):
    """ReflectionPad2d((K-1)//2) + Conv2d [+ ReLU] [+ out = old + sign * conv] (models/RevResNet.py:79-88,96-116)."""
    B, Cin, H, W = x.shape
    Cout, _, K, _ = weight.shape
    pad = (K - 1) // 2
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    if out is None:
        out = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().vst_generic_conv(_p(x), _p(weight), _p(bias), _p(old), float(sign), int(relu), _p(out), B, Cin, Cout, H, W,
                                           K, stride, _st()), "vst_generic_conv")
    return out


def squeeze(x):
    """models/RevResNet.py:34-37."""
    B, D, H, W = x.shape
    y = torch.empty((B, 4 * D, H // 2, W // 2), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().vst_generic_squeeze(_p(x), _p(y), B, D, H, W, _st()), "vst_generic_squeeze")
    return y


def unsqueeze(y):
    """models/RevResNet.py:40-43."""
    B, C4, h, w = y.shape
    x = torch.empty((B, C4 // 4, 2 * h, 2 * w), dtype=torch.float32, device=y.device)
    _lib.check(_lib.lib().vst_generic_unsqueeze(_p(y), _p(x), B, C4 // 4, 2 * h, 2 * w, _st()), "vst_generic_unsqueeze")
    return x


def channels(src, c0, n, out=None, d0=0, c_out=None):
    """out[:, d0:d0+n] = src[:, c0:c0+n] (split / merge / injective_pad as channel-range copies, models/RevResNet.py:8-31);
    a fresh `out` with more channels than it receives is zero-filled first (inj_pad.forward)."""
    B, Cs, H, W = src.shape
    if out is None:
        c_out = n if c_out is None else c_out
        out = torch.empty((B, c_out, H, W), dtype=torch.float32, device=src.device)
        if c_out > n:
            _lib.check(_lib.lib().vst_generic_zero(_p(out), out.numel(), _st()), "vst_generic_zero")
    _lib.check(_lib.lib().vst_generic_copy_channels(_p(src), _p(out), B, Cs, c0, n, H * W, out.shape[1], d0, _st()),
               "vst_generic_copy_channels")
    return out


def _convs(block):
    return block.conv[1], block.conv[4], block.conv[7]


def residual_F(block, x2, old=None, sign=1.0):
    """F(x2) of models/RevResNet.py:79-88; with `old`: old + sign * F(x2) in the last conv's epilogue."""
    c1, c4, c7 = _convs(block)
    h = conv(x2, c1.weight, c1.bias, stride=block.stride, relu=True)
    h = conv(h, c4.weight, c4.bias, relu=True)
    return conv(h, c7.weight, c7.bias, old=old, sign=sign)


def block_forward(block, x1, x2):
    """models/RevResNet.py:96-104: (x1, x2) -> (x2', F(x2) + x1')."""
    if block.stride == 2:
        x1s, x2s = squeeze(x1), squeeze(x2)
        return x2s, residual_F(block, x2, old=x1s, sign=1.0)
    return x2, residual_F(block, x2, old=x1, sign=1.0)


def block_inverse(block, x2, y1):
    """models/RevResNet.py:106-116: (x2', y1) -> (x1, x2)."""
    if block.stride == 2:
        x2 = unsqueeze(x2)
    x1 = residual_F(block, x2, old=y1, sign=-1.0)
    if block.stride == 2:
        x1 = unsqueeze(x1)
    return x1, x2


def forward(net, x):
    """models/RevResNet.py:210-223 + channel_reduction.forward :131-146."""
    c0 = net.in_ch
    x = channels(x, 0, x.shape[1], c_out=2 * c0)                       # inj_pad.forward
    x1, x2 = channels(x, 0, c0), channels(x, c0, c0)                   # split
    for blk in net.stack:
        x1, x2 = block_forward(blk, x1, x2)
    cr = net.channel_reduction
    full = x1.shape[1] + cr.pad                                       # each half is padded to out_ch * 4**sp_steps channels
    if cr.pad:
        x1, x2 = channels(x1, 0, x1.shape[1], c_out=full), channels(x2, 0, x2.shape[1], c_out=full)
    for blk in cr.block_list:
        x1, x2 = block_forward(blk, x1, x2)
    z = channels(x1, 0, full, c_out=2 * full)                          # merge
    channels(x2, 0, full, out=z, d0=full)
    for _ in range(cr.sp_steps):                                       # "spread" == unsqueeze
        z = unsqueeze(z)
    return z


def inverse(net, z):
    """models/RevResNet.py:225-239 + channel_reduction.inverse :148-163."""
    cr = net.channel_reduction
    for _ in range(cr.sp_steps):
        z = squeeze(z)
    full = z.shape[1] // 2
    a, b = channels(z, 0, full), channels(z, full, full)
    for blk in list(cr.block_list)[::-1]:
        a, b = block_inverse(blk, a, b)
    if cr.pad:                                                         # inj_pad.inverse of both halves: drop the padded channels
        a, b = channels(a, 0, full - cr.pad), channels(b, 0, full - cr.pad)
    for blk in list(net.stack)[::-1]:
        a, b = block_inverse(blk, a, b)
    c0 = net.in_ch
    return channels(a, 0, net.in_channel) if net.in_channel <= c0 else _merge_drop(a, b, net)


def _merge_drop(a, b, net):
    """merge + inj_pad.inverse when the image has more channels than one half (in_channel > nChannels[0])."""
    c0 = net.in_ch
    m = channels(a, 0, c0, c_out=2 * c0)
    channels(b, 0, c0, out=m, d0=c0)
    return channels(m, 0, net.in_channel)
