"""Export a RevResNet state_dict to the flat file the native runner (tools/vst_run.cpp) reads.

Layout: magic "VSTW", int32 version(1), int32 hidden_dim, int32 sp_steps, then the 192 tensors of the reference's
state_dict in its own order (stack.{i}.conv.{1,4,7}.{weight,bias}, channel_reduction.block_list.{0,1}...), raw
little-endian fp32, shapes implied by the architecture (models/RevResNet.py:68-94,166-201)."""
import struct

import numpy as np

from .synth import state_dict_spec


def export_state_dict(state_dict, path, hidden_dim=16, sp_steps=2):
    with open(path, "wb") as f:
        f.write(b"VSTW")
        f.write(struct.pack("<iii", 1, hidden_dim, sp_steps))
        for key, shape in state_dict_spec(hidden_dim, sp_steps):
            t = state_dict[key].detach().cpu().float().contiguous().numpy()
            assert tuple(t.shape) == tuple(shape), (key, t.shape, shape)
            f.write(np.ascontiguousarray(t, dtype="<f4").tobytes())
    return path
