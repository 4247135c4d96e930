"""Test infrastructure: the hostile checkpoints and error measures of the robustness tests (tests/test_gpu_robust.py on the
card, tests/test_robust_model.py for the CPU model of the same cases).

The reference's trained checkpoints are not available offline, so the regular parity tests run on `synthetic_state_dict`
(O(1), well-conditioned codes: per-channel std of z 0.07 .. 0.4, cond(cov z) ~ 8e2).  These helpers derive checkpoints from it
whose codes are NOT friendly, with the same key / shape contract (models/RevResNet.py:68-94,119-129):

  ramp_state_dict      conv.7 of channel_reduction.block_list.{0,1} scaled per output channel by a geometric ramp over the
                       code channel it lands in (state channel c of either half is code channel c mod N): per-channel std of z
                       spans >= `span`, cond(cov z) >= 1e5 at span 1e3 — the low-variance channels are the ones whitening
                       (models/cWCT.py:134-149) amplifies.
  rescale_state_dict   a function-preserving rescaling of the intermediates: h1 channels x f1, h2 channels x f2 (ReLU is
                       positively homogeneous: W1,b1 *= f1; W4 *= f2/f1; b4 *= f2; W7 /= f2).  The reference computes the same
                       function (to fp32 rounding); an fp16 operand path sees h1 / h2 a factor f1 / f2 away from O(1).
"""
from __future__ import annotations

import numpy as np
import torch

BLOCK_PREFIXES = [f"stack.{i}." for i in range(30)] + [f"channel_reduction.block_list.{i}." for i in range(2)]


def ramp_state_dict(sd, span, n_code):
    sd = {k: v.clone() for k, v in sd.items()}
    r = torch.tensor(span ** (np.arange(256) % n_code / (n_code - 1)), dtype=torch.float32)
    for i in range(2):
        p = f"channel_reduction.block_list.{i}.conv.7."
        sd[p + "weight"] *= r[:, None, None, None]
        sd[p + "bias"] *= r
    return sd


def rescale_state_dict(sd, f1, f2):
    sd = {k: v.clone() for k, v in sd.items()}
    for p in BLOCK_PREFIXES:
        sd[p + "conv.1.weight"] *= f1
        sd[p + "conv.1.bias"] *= f1
        sd[p + "conv.4.weight"] *= f2 / f1
        sd[p + "conv.4.bias"] *= f2
        sd[p + "conv.7.weight"] /= f2
    return sd


def code_conditioning(z):
    """(min, max) of the per-channel std and cond(cov) of one image's code [1,N,h,w] (fp64)."""
    m = z[0].reshape(z.shape[1], -1).double()
    std = m.std(dim=1)
    mc = m - m.mean(1, keepdim=True)
    cov = mc @ mc.t() / (m.shape[1] - 1)
    return float(std.min()), float(std.max()), float(torch.linalg.cond(cov))


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def max_rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def worst_channel(a, b):
    """max over channels c of ||a_c - b_c|| / ||b_c||: the error measure a per-tensor norm hides when the channels'
    scales differ by orders of magnitude."""
    a, b = a.double().cpu(), b.double().cpu()
    N = a.shape[1]
    d = (a - b).transpose(0, 1).reshape(N, -1).norm(dim=1)
    r = b.transpose(0, 1).reshape(N, -1).norm(dim=1)
    return float((d / (r + 1e-300)).max())


def normalized_state_dict(sd):
    """CPU restatement of vst_normalize_block (include/vstnet.h) for the CPU model of the fp16 modes (tests/emul.py):
    s = 2^-round(log2 ||row||_2) per intermediate channel, applied function-preservingly."""
    sd = {k: v.clone() for k, v in sd.items()}
    for p in BLOCK_PREFIXES:
        w1, w4, w7 = sd[p + "conv.1.weight"], sd[p + "conv.4.weight"], sd[p + "conv.7.weight"]
        s1 = torch.exp2(-torch.round(torch.log2(w1.flatten(1).norm(dim=1).clamp_min(1e-30))))
        w1 *= s1[:, None, None, None]
        sd[p + "conv.1.bias"] *= s1
        w4 /= s1[None, :, None, None]
        s2 = torch.exp2(-torch.round(torch.log2(w4.flatten(1).norm(dim=1).clamp_min(1e-30))))
        w4 *= s2[:, None, None, None]
        sd[p + "conv.4.bias"] *= s2
        w7 /= s2[None, :, None, None]
    return sd
