"""Test infrastructure: a CPU *model* of the HIP path's narrowed arithmetic modes, built on the oracle.

Used to design and price the robustness cases of tests/test_gpu_robust.py on the CPU before they run on the card
(ill-conditioned codes, scaled inputs / weights).  It models WHERE each mode rounds, not the MFMA summation order:

  f16x2  : weights rounded to fp16 once; every conv input kept as fp16 hi + lo (22 bits, absolute floor 2^-25).
  f16x2h : in addition h1 / h2 (the inputs of conv.4 / conv.7) are single fp16 values and the 256-channel blocks' conv.1
           reads only the hi plane of the state; the stage-3 state lives in hi + lo planes.
  bf16x3 : a_hi w_hi + a_lo w_hi + a_hi w_lo with bf16 hi / lo: modelled as 16-bit operands.

Not product code; imports the oracle (allowed under tests/ only).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from oracle import cpu_ref


def f16(x):
    return x.clamp(-65504.0, 65504.0).half().float()


def f16x2(x):
    hi = f16(x)
    return hi + (x - hi).half().float()


def bf16x2(x):
    hi = x.bfloat16().float()
    return hi + (x - hi).bfloat16().float()


def _conv(x, w, b, stride=1):
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b, stride=stride)


def residual_F(x2, sd, prefix, stride, mode, channel):
    w1, w4, w7 = (sd[prefix + f"conv.{i}.weight"] for i in (1, 4, 7))
    b1, b4, b7 = (sd[prefix + f"conv.{i}.bias"] for i in (1, 4, 7))
    if mode == "bf16x3":
        q = bf16x2
        h = F.relu(_conv(q(x2), q(w1), b1, stride))
        h = F.relu(_conv(q(h), q(w4), b4))
        return _conv(q(h), q(w7), b7)
    w1, w4, w7 = f16(w1), f16(w4), f16(w7)
    narrow = mode == "f16x2h"
    x_in = f16(x2) if (narrow and channel == 256 and stride == 1) else f16x2(x2)
    h = F.relu(_conv(x_in, w1, b1, stride))
    h = F.relu(_conv(f16(h) if narrow else f16x2(h), w4, b4))
    return _conv(f16(h) if narrow else f16x2(h), w7, b7)


def revnet_forward(x, sd, sp_steps, mode):
    x = cpu_ref.inj_pad_fwd(x, 32 - x.shape[1])
    x1, x2 = cpu_ref.split(x)
    for i, (stride, ch) in enumerate(cpu_ref.STACK):
        fx = residual_F(x2, sd, f"stack.{i}.", stride, mode, ch)
        if stride == 2:
            x1, x2 = cpu_ref.squeeze(x1), cpu_ref.squeeze(x2)
        x1, x2 = x2, fx + x1
        if ch == 256 and mode != "bf16x3":
            x2 = f16x2(x2)                     # the stage-3 state lives in hi + lo planes
    for i in range(2):
        fx = residual_F(x2, sd, f"channel_reduction.block_list.{i}.", 1, mode, 256)
        x1, x2 = x2, fx + x1
    z = cpu_ref.merge(x1, x2)
    for _ in range(sp_steps):
        z = cpu_ref.unsqueeze(z)
    return z


def revnet_inverse(z, sd, sp_steps, mode, in_channel=3):
    for _ in range(sp_steps):
        z = cpu_ref.squeeze(z)
    a, b = cpu_ref.split(z)
    for i in (1, 0):
        x2 = a
        if mode != "bf16x3":
            x2 = f16x2(x2)
        a, b = b - residual_F(x2, sd, f"channel_reduction.block_list.{i}.", 1, mode, 256), x2
    for i in range(len(cpu_ref.STACK) - 1, -1, -1):
        stride, ch = cpu_ref.STACK[i]
        x2 = a
        if stride == 2:
            x2 = cpu_ref.unsqueeze(x2)
        x1 = b - residual_F(x2, sd, f"stack.{i}.", stride, mode, ch)
        if stride == 2:
            x1 = cpu_ref.unsqueeze(x1)
        a, b = x1, x2
    return cpu_ref.inj_pad_inv(cpu_ref.merge(a, b), 32 - in_channel)


def stylize(xc, xs, sd, sp_steps, mode):
    zc, zs = revnet_forward(xc, sd, sp_steps, mode), revnet_forward(xs, sd, sp_steps, mode)
    zcs = cpu_ref.transfer(zc, zs)
    return zc, zs, zcs, revnet_inverse(zcs, sd, sp_steps, mode)
