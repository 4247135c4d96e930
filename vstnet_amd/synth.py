"""Deterministic synthetic checkpoints and frames.

The reference's trained checkpoints (checkpoints/photo_image.pt, ...) are not
available offline, so every parity test, the smoke test and bench.py run on a
synthetic ``state_dict`` with exactly the reference's key names and shapes
(reference: models/RevResNet.py:68-94 residual_block, :119-129
channel_reduction, :166-201 RevResNet; SURVEY.md section 8(b) "State-dict
contract").  The generator is numpy-only so that the values do not depend on
the torch version, and biases are non-zero (the reference's default init zeroes
them, models/RevResNet.py:91-94, which would hide bias bugs).
"""
from __future__ import annotations

import numpy as np
import torch

# (stride, channel) of the 30 blocks of the stack, reference models/RevResNet.py:192-201
STACK = [(1, 16)] * 10 + [(2, 64)] + [(1, 64)] * 9 + [(2, 256)] + [(1, 256)] * 9
CONV_IDX = (1, 4, 7)  # positions of the Conv2d modules inside residual_block.conv


def block_conv_shapes(channel: int, stride: int, mult: int = 4):
    """OIHW shapes of the three convs of one residual_block (models/RevResNet.py:72-88)."""
    in_ch = channel if stride == 1 else channel // 4
    mid = channel // mult
    return [(mid, in_ch, 3, 3), (mid, mid, 3, 3), (channel, mid, 3, 3)]


def state_dict_spec(hidden_dim: int = 16, sp_steps: int = 2):
    """Ordered list of (key, shape) pairs — 192 tensors for both modes."""
    spec = []
    for i, (stride, ch) in enumerate(STACK):
        for ci, shp in zip(CONV_IDX, block_conv_shapes(ch, stride)):
            spec.append((f"stack.{i}.conv.{ci}.weight", shp))
            spec.append((f"stack.{i}.conv.{ci}.bias", (shp[0],)))
    cr_ch = hidden_dim * 4 ** sp_steps
    for i in range(2):
        for ci, shp in zip(CONV_IDX, block_conv_shapes(cr_ch, 1)):
            spec.append((f"channel_reduction.block_list.{i}.conv.{ci}.weight", shp))
            spec.append((f"channel_reduction.block_list.{i}.conv.{ci}.bias", (shp[0],)))
    return spec


def synthetic_state_dict(seed: int = 1234, hidden_dim: int = 16, sp_steps: int = 2,
                         weight_gain: float = 1.0, bias_scale: float = 0.05):
    """Seeded fp32 state_dict.  Weights ~ U(-g/sqrt(fan_in), g/sqrt(fan_in)), biases ~ U(-b, b)."""
    out = {}
    for idx, (key, shp) in enumerate(state_dict_spec(hidden_dim, sp_steps)):
        rng = np.random.Generator(np.random.PCG64([seed, idx]))
        if len(shp) == 4:
            bound = weight_gain / np.sqrt(shp[1] * shp[2] * shp[3])
        else:
            bound = bias_scale
        arr = rng.uniform(-bound, bound, size=shp).astype(np.float32)
        out[key] = torch.from_numpy(arr)
    return out


def synthetic_frames(batch: int, height: int, width: int, seed: int = 0) -> torch.Tensor:
    """[B,3,H,W] fp32 in [0,1): frame f is generated from (seed, f) so shards can make their own."""
    frames = []
    for f in range(batch):
        rng = np.random.Generator(np.random.PCG64([seed, f]))
        frames.append(rng.random((3, height, width), dtype=np.float32))
    return torch.from_numpy(np.stack(frames))


def synthetic_mask(height: int, width: int, labels: int = 5, seed: int = 0, speck: bool = True,
                   kind: str = "bands") -> np.ndarray:
    """uint8 [H,W] label map (+ one <=10-px speck of label `labels` that exercises the validity rule of
    models/cWCT.py:178).  kind "bands": vertical bands of `labels` labels; kind "noise": an independent uniform
    label per pixel — every 64-pixel tile holds every label, the worst case for per-label passes."""
    rng = np.random.Generator(np.random.PCG64([seed, 77]))
    if kind == "noise":
        m = rng.integers(0, labels, size=(height, width), dtype=np.uint8)
        if speck:
            m[1:3, 1:4] = labels
        return m
    if kind != "bands":
        raise ValueError("kind must be 'bands' or 'noise'")
    cuts = np.sort(rng.choice(np.arange(8, width - 8), size=labels - 1, replace=False))
    m = np.zeros((height, width), dtype=np.uint8)
    for i, c in enumerate(cuts):
        m[:, c:] = i + 1
    if speck:
        m[1:3, 1:4] = labels  # 6 pixels -> invalid label
    return m
