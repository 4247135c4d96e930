"""Test helpers: NCHW <-> ZC state layout conversions written with torch ops (independent of the HIP
kernels), and raw C-ABI call wrappers."""
import ctypes as C

import torch

from oracle import cpu_ref


def nchw_to_zc(x):
    """x: [B,C,h,w] at view level l (C = 16*4**l) -> ZC state [B,H/4,W/4,256]."""
    while x.shape[1] < 256:
        x = cpu_ref.squeeze(x)
    return x.permute(0, 2, 3, 1).contiguous()


def zc_to_nchw(state, channels):
    """ZC state [B,Hq,Wq,256] -> [B,channels,h,w] view."""
    x = state.permute(0, 3, 1, 2).contiguous()
    while x.shape[1] > channels:
        x = cpu_ref.unsqueeze(x)
    return x


def ptr(t):
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    l2 = float((a - b).norm() / (b.norm() + 1e-30))
    mx = float((a - b).abs().max() / (b.abs().max() + 1e-30))
    return l2, mx
