"""PackedCode: the photorealistic code z = RevResNet(x) kept in the layout the coupling blocks leave it in.

The reference's forward pass ends with merge + two unsqueeze steps (models/RevResNet.py:139-144, :219-222) and its inverse
begins by undoing them (:148-154, :228-231): pure (channel, pixel) permutations, 2 x 256 MB of HBM traffic per 1024x1024
frame.  An unmasked cWCT (models/cWCT.py:24-47, :206-262) does not depend on the order of the pixels, so on the video path
encode -> cWCT -> decode nobody needs z in NCHW order.  ``net(x)`` therefore returns a PackedCode: a tensor that *is*
``[B, 32, H, W]`` float32 to every caller (shape, dtype, device; any torch operation on it first materialises the NCHW
values, once, with the library's spread kernel), while ``cWCT.transfer / interpolation / transfer_with_stats`` and
``net(z, forward=False)`` recognise it and work on the packed rows directly (include/vstnet.h, "Packed code").
A cWCT result is a PackedCode with a pending affine map per image, applied while the inverse pass loads its state.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch.utils._pytree import tree_map

from . import _lib


def _stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class PackedCode(torch.Tensor):
    """float[B][2][H/4][W/4][256] behind the interface of the [B,32,H,W] (or [B,128,H/2,W/2]) code tensor."""

    @staticmethod
    def __new__(cls, code, H, W, affines=None, labels=None, sp_steps=2):
        B = code.shape[0]
        shape = (B, 32, H, W) if sp_steps == 2 else (B, 128, H // 2, W // 2)
        r = torch.Tensor._make_wrapper_subclass(cls, shape, dtype=torch.float32, device=code.device, requires_grad=False)
        r._code = code            # flat float32 [B, 32*H*W], packed rows
        r._hw = (H, W)            # of the IMAGE (the code of an artistic net is [B,128,H/2,W/2])
        r._sp = sp_steps          # 2: rows of 32 (photorealistic), 1: rows of 128 (artistic)
        r._affines = affines      # None, or float32 [B, N*N+N]: y = T x + t0 still to be applied to every row of image b
        # None, or the pending MASKED cWCT: per image (affines of its label slots [slots, 1056], labels of its rows uint8
        # [H*W], its label plan), and the slot count the launches cover
        r._labels = labels
        r._dense = None
        r._version0 = r._version  # of this wrapper: torch bumps it on every in-place op on the code or on a view of it
        return r

    def __repr__(self):
        H, W = self._hw
        what = "affine" if self._affines is not None else ("masked" if self._labels is not None else "none")
        return f"PackedCode(B={self._code.shape[0]}, N={self.shape[1]}, image {H}x{W}, pending={what})"

    @property
    def packed(self):
        return self._code

    @property
    def image_hw(self):
        return self._hw

    @property
    def sp_steps(self):
        return self._sp

    @property
    def pending_affines(self):
        return self._affines

    @property
    def pending_labels(self):
        return self._labels

    @property
    def pending(self):
        return self._affines is not None or self._labels is not None

    @property
    def stale(self):
        """True once somebody has WRITTEN to the code (z.mul_(2), z[:, :, a:b] = v, v = z[:, 1]; v.add_(1), out=z ...).  Such
        edits land in the materialised dense tensor (__torch_dispatch__ runs the op on it), and torch's in-place / view
        tracking bumps THIS tensor's version counter for them — also for writes through views, which share it — so the packed
        rows no longer are the code.  Every consumer of the rows (net(z, forward=False), inverse_u8, the cWCT packed routes)
        then takes the dense values instead, like for any plain tensor."""
        return self._version != self._version0

    def _need_gpu(self):
        if not self._code.is_cuda:
            raise RuntimeError("vstnet_amd.PackedCode lives on ROCm devices only (no CPU fallback)")

    def with_affines(self, affines):
        """The same packed rows with the affine map of a cWCT attached (composition is not supported: materialise first)."""
        assert not self.pending
        H, W = self._hw
        return PackedCode(self._code, H, W, affines, None, self._sp)

    def with_label_affines(self, per_image, max_slots):
        """The same rows with a masked cWCT attached: per_image[b] = (affines [slots,1056], row labels, label plan)."""
        assert not self.pending
        H, W = self._hw
        assert self._sp == 2
        return PackedCode(self._code, H, W, None, (list(per_image), int(max_slots)))

    def applied(self):
        """Packed rows with the pending affine map applied (a new buffer), or the rows themselves if none is pending."""
        if not self.pending:
            return self._code
        self._need_gpu()
        L = _lib.lib()
        H, W = self._hw
        out = torch.empty_like(self._code)
        with torch.cuda.device(self._code.device):
            for b in range(self._code.shape[0]):
                if self._affines is not None:
                    _lib.check(L.vst_cwct_apply_code(C.c_void_p(self._code[b].data_ptr()), C.c_void_p(out[b].data_ptr()), H, W,
                                                     self._sp, C.c_void_p(self._affines[b].data_ptr()), _stream_ptr()),
                               "vst_cwct_apply_code")
                else:
                    aff, rows, plan = self._labels[0][b]
                    _lib.check(L.vst_cwct_apply_labels_code(C.c_void_p(self._code[b].data_ptr()), C.c_void_p(out[b].data_ptr()),
                                                            H, W, C.c_void_p(aff.data_ptr()), C.c_void_p(rows.data_ptr()),
                                                            C.c_void_p(plan.data_ptr()), self._labels[1], _stream_ptr()),
                               "vst_cwct_apply_labels_code")
        return out

    def materialize(self):
        """The [B,32,H,W] values (NCHW, a plain tensor), computed once."""
        if self._dense is None:
            self._need_gpu()
            L = _lib.lib()
            H, W = self._hw
            rows = self.applied()
            z = torch.empty(tuple(self.shape), dtype=torch.float32, device=rows.device)
            with torch.cuda.device(rows.device):
                _lib.check(L.vst_code_to_z(C.c_void_p(rows.data_ptr()), C.c_void_p(z.data_ptr()), rows.shape[0], H, W,
                                           self._sp, _stream_ptr()), "vst_code_to_z")
            self._dense = z
        return self._dense

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        def unwrap(t):
            return t.materialize() if isinstance(t, PackedCode) else t
        return func(*tree_map(unwrap, args), **tree_map(unwrap, kwargs or {}))


def from_dense(z):
    """Pack a plain [B,32,H,W] or [B,128,H/2,W/2] code (vst_z_to_code)."""
    L = _lib.lib()
    z = z.detach().to(torch.float32).contiguous()
    B, N = z.shape[:2]
    sp = 2 if N == 32 else 1
    H, W = (z.shape[2], z.shape[3]) if sp == 2 else (2 * z.shape[2], 2 * z.shape[3])
    code = torch.empty((B, 32 * H * W), dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        _lib.check(L.vst_z_to_code(C.c_void_p(z.data_ptr()), C.c_void_p(code.data_ptr()), B, H, W, sp, _stream_ptr()),
                   "vst_z_to_code")
    return PackedCode(code, H, W, None, None, sp)
