"""cWCT — drop-in for the reference's ``models.cWCT.cWCT`` on MI355X.

Same constructor and methods as models/cWCT.py:9-262 (``transfer``, ``interpolation`` and the
public-by-convention helpers).  Differences, all documented in DESIGN.md:
  * ``transfer`` without masks implements the intended per-sample semantics (the fork's batched
    ``whitening`` raises on [B,N,L] input);
  * the Cholesky-failure path retries with the reference's jitter schedule but never enters pdb;
  * masks must have the feature resolution (the fork does not resize them, cWCT.py:72-73): a
    mismatch raises ValueError instead of indexing out of range;
  * ``use_double=True`` (cWCT.py:13-16,35-47,66,106,220,238,259; no script of the reference sets it, and in this fork the flag
    traps into pdb, :36) runs true fp64 two-pass statistics, an fp64 Cholesky / inverse / mix and an fp64-accumulating apply
    (csrc/cwct64.hip) on the dense NCHW code: a fidelity option — packed codes are materialised first, masked transfers go
    label by label like the reference's loop.  Without it the mean / covariance combine is fp64, the rest fp32;
  * a batch is factored like the reference's [B,N,N] stack: a sample that needs Cholesky jitter jitters every sample
    (cWCT.py:122-128); ``transfer_with_stats`` / ``transfer_with_plan`` (this repo's cached-style extensions) are per sample.
All device work goes through libvstnet_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .code import PackedCode

_SUPPORTED_N = (16, 32, 64, 128)


def _stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class MaskPlan:
    """What the masked transfer derives from the label maps alone (cWCT.plan_masks): per sample the uint8 maps and the
    label -> slot table, all on the device; optionally the per-slot style statistics (cWCT.bind_style)."""

    def __init__(self):
        self.cm, self.sm, self.tables, self.shapes, self.style, self.max_slots = [], [], [], None, None, 0
        self.cm_rows = None        # the content label maps in a PackedCode's row order (made on first use)


class cWCT(nn.Module):
    """Cholesky decomposition based WCT (HIP implementation)."""

    def __init__(self, eps=2e-5, use_double=False, resize_masks=False, precision=None):
        super().__init__()
        # arithmetic of the apply (y = T x + t0): "fp32" = exact fp32 kernels for every N; anything else lets unmasked
        # N >= 64 codes (artistic mode) run on bf16 MFMA with split operands.  Follows RevResNet's knob by default.
        precision = precision or _lib.default_precision()
        if precision == "auto":               # (RevResNet's self-calibrating mode: nothing to decide here)
            precision = "bf16x3"
        if precision not in ("fp32", "bf16x3", "f16x2", "f16x2h"):
            raise ValueError("precision must be one of ['auto', 'bf16x3', 'f16x2', 'f16x2h', 'fp32']")
        self.precision = precision
        self.eps = eps
        self.use_double = bool(use_double)
        # upstream CAP-VSTNet resized the label maps to the feature resolution (NEAREST, cWCT.py:191-197); this
        # fork uses them as they are (:72-73), which only fits photorealistic codes.  Opt in to restore it.
        self.resize_masks = resize_masks
        self._ws = None
        self.last_info = None      # device int32 [2+n_styles]: content retries, overflow flag, style retries
        self.last_route = None     # key of ROUTES the last transfer took

    # ------------------------------------------------------------------ low-level wrappers
    def _workspace(self, nbytes, device):
        """Statistics workspace, one per (device, stream)."""
        if self._ws is None:
            self._ws = {}
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            self._ws[key] = ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        return ws

    @staticmethod
    def _prep(x):
        if not x.is_cuda:
            raise RuntimeError("vstnet_amd.cWCT runs on ROCm devices only (no CPU fallback)")
        return x.detach().to(torch.float32).contiguous()

    def stats(self, x2d, mask=None, label=0):
        """mean / covariance of a [N,L] feature matrix (optionally of the pixels with mask==label)
        -> device double tensor [1+N+N*N] = {n, mean, cov}  (cWCT.py:138-144 / 153-157)."""
        N, Lp = x2d.shape
        if N not in _SUPPORTED_N:
            raise NotImplementedError(f"HIP cWCT supports N in {_SUPPORTED_N}, got {N}")
        L = _lib.lib()
        out = torch.empty(1 + N + N * N, dtype=torch.float64, device=x2d.device)
        if self.use_double:
            ws = self._workspace(L.vst_cwct_stats_f64_workspace_bytes(N, Lp), x2d.device)
            with torch.cuda.device(x2d.device):
                _lib.check(L.vst_cwct_stats_f64(_ptr(x2d), N, Lp, _ptr(mask), int(label), _ptr(out), _ptr(ws), _stream_ptr()),
                           "vst_cwct_stats_f64")
            return out
        ws = self._workspace(L.vst_cwct_stats_workspace_bytes(N, Lp), x2d.device)
        with torch.cuda.device(x2d.device):
            _lib.check(L.vst_cwct_stats(_ptr(x2d), N, Lp, _ptr(mask), int(label), _ptr(out), _ptr(ws), _stream_ptr()),
                       "vst_cwct_stats")
        return out

    def stats_code(self, z, b):
        """stats() of image b of a PackedCode (all pixels), on the packed rows (vst_cwct_stats_code)."""
        rows = z.applied()[b]
        H, W = z.image_hw
        N = z.shape[1]
        L = _lib.lib()
        out = torch.empty(1 + N + N * N, dtype=torch.float64, device=rows.device)
        ws = self._workspace(L.vst_cwct_stats_code_workspace_bytes(H, W, z.sp_steps), rows.device)
        with torch.cuda.device(rows.device):
            _lib.check(L.vst_cwct_stats_code(_ptr(rows), H, W, z.sp_steps, _ptr(out), _ptr(ws), _stream_ptr()),
                       "vst_cwct_stats_code")
        return out

    # ------------------------------------------------------------------ the ONE place where a transfer's route is chosen
    # (code layout x mask x N x slot count x use_double) -> which kernels run.  The arithmetic of the dense applies (exact fp32
    # vs bf16 split operands for N >= 64) is the `precision` argument of vst_cwct_apply_prec / vst_cwct_apply_labels and is
    # chosen inside the library; everything else is decided here and recorded in `last_route`.
    ROUTES = {
        "packed_rows": "unmasked, code in the coupling blocks' layout: vst_cwct_stats_code + factor; the map stays pending and "
                       "is applied by the inverse pass (vst_revnet_decode)",
        "dense": "unmasked NCHW code: vst_cwct_stats + factor + vst_cwct_apply_prec",
        "masked_packed_rows": "masked, photorealistic packed code, 1..8 label slots known: vst_cwct_stats_labels_code + "
                              "factor_labels; per-row maps pending (vst_revnet_decode_labels)",
        "masked_single_pass": "masked NCHW code, N in {32, 64, 128}: vst_cwct_stats_labels + factor_labels + apply_labels",
        "masked_per_label": "masked NCHW code, N = 16: one vst_cwct_stats / factor / apply per valid label",
        "dense_f64": "use_double, unmasked: vst_cwct_stats_f64 + factor_f64 + apply_f64 on the NCHW code",
        "masked_per_label_f64": "use_double, masked: the fp64 calls per valid label (the reference's loop, cWCT.py:83-103)",
    }

    @staticmethod
    def route(packed, masked, N, sp_steps=2, max_slots=0, use_double=False):
        """Name of the route (a key of ROUTES) for a code that is / is not a usable PackedCode (`packed`: no pending map, not
        written to), with / without masks, N channels, `max_slots` label slots known to the host (0 = never read back)."""
        if N not in _SUPPORTED_N:
            raise NotImplementedError(f"HIP cWCT supports N in {_SUPPORTED_N}, got {N}")
        if use_double:
            return "masked_per_label_f64" if masked else "dense_f64"
        if not masked:
            return "packed_rows" if packed and ((N == 32 and sp_steps == 2) or (N == 128 and sp_steps == 1)) else "dense"
        if N == 16:
            return "masked_per_label"
        if packed and N == 32 and sp_steps == 2 and 1 <= int(max_slots) <= 8:
            return "masked_packed_rows"
        return "masked_single_pass"

    def _route_of(self, content_feat, masked, max_slots=0):
        r = self.route(self._is_packed_code(content_feat), masked, content_feat.shape[1],
                       getattr(content_feat, "sp_steps", 2), max_slots, self.use_double)
        self.last_route = r
        return r

    @staticmethod
    def _is_packed_code(x):
        """A code still in the coupling blocks' layout, no cWCT pending on it (code.py); use_double works on dense codes."""
        return isinstance(x, PackedCode) and not x.pending and not x.stale

    def factor(self, content_stats, style_stats_list, alphas, alpha_c, N, min_tries=None):
        """{T, t0} with T = (sum_i a_i chol(Cs_i) [blended with chol(Cc)]) * chol(Cc)^-1.  min_tries: device int32
        [2+n_styles] jitter retries to start from (the batch coupling of `interpolation`)."""
        L = _lib.lib()
        n = len(style_stats_list)
        dev = content_stats.device
        info = torch.zeros(2 + n, dtype=torch.int32, device=dev) if min_tries is None else min_tries.clone()
        ptrs = (C.c_void_p * n)(*[s.data_ptr() for s in style_stats_list])
        al = (C.c_float * n)(*[float(a) for a in alphas])
        if self.use_double:                     # fp64 Cholesky / inverse / mix: a DOUBLE affine record
            affine = torch.empty(N * N + N, dtype=torch.float64, device=dev)
            fws = torch.empty(L.vst_cwct_factor_f64_workspace_bytes(N), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _lib.check(L.vst_cwct_factor_f64(_ptr(content_stats), ptrs, al, n, float(alpha_c), float(self.eps), N,
                                                 _ptr(affine), _ptr(info), _ptr(fws), _stream_ptr()), "vst_cwct_factor_f64")
            self.last_info = info
            return affine
        affine = torch.empty(N * N + N, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.vst_cwct_factor(_ptr(content_stats), ptrs, al, n, float(alpha_c), float(self.eps), N,
                                         _ptr(affine), _ptr(info), _stream_ptr()), "vst_cwct_factor")
        self.last_info = info
        return affine

    def apply(self, x2d, affine, out=None, mask=None, label=0):
        N, Lp = x2d.shape
        if out is None:
            out = torch.empty_like(x2d)
        if self.use_double:
            with torch.cuda.device(x2d.device):
                _lib.check(_lib.lib().vst_cwct_apply_f64(_ptr(x2d), _ptr(out), N, Lp, _ptr(affine), _ptr(mask), int(label),
                                                         _stream_ptr()), "vst_cwct_apply_f64")
            return out
        with torch.cuda.device(x2d.device):
            prec = {"fp32": _lib.PREC_FP32, "bf16x3": _lib.PREC_BF16X3, "f16x2": _lib.PREC_F16X2, "f16x2h": _lib.PREC_F16X2H}[self.precision]
            _lib.check(_lib.lib().vst_cwct_apply_prec(_ptr(x2d), _ptr(out), N, Lp, _ptr(affine), _ptr(mask), int(label),
                                                      prec, _stream_ptr()), "vst_cwct_apply_prec")
        return out

    @staticmethod
    def _identity_stats(N, device):
        s = torch.zeros(1 + N + N * N, dtype=torch.float64, device=device)
        s[0] = 2.0
        s[1 + N:] = torch.eye(N, dtype=torch.float64, device=device).reshape(-1)
        return s

    # ------------------------------------------------------------------ reference surface
    def transfer(self, content_feat, style_feat, cmask=None, smask=None):
        """models/cWCT.py:18-22."""
        if cmask is None or smask is None:
            return self._transfer(content_feat, style_feat)
        return self._transfer_seg(content_feat, style_feat, cmask, smask)

    def _transfer(self, content_feat, style_feat):
        """models/cWCT.py:24-47, per sample (== interpolation(c,[s],[1.0],0.0))."""
        return self.interpolation(content_feat, [style_feat], [1.0], 0.0)

    def interpolation(self, content_feat, styl_feat_list, alpha_s_list, alpha_c=0.0):
        """models/cWCT.py:206-262."""
        assert len(styl_feat_list) == len(alpha_s_list)
        B, N, cH, cW = content_feat.shape
        in_dtype = content_feat.dtype
        packed = self._route_of(content_feat, masked=False) == "packed_rows"   # statistics on the packed rows; map applied by the inverse pass
        c = None if packed else self._prep(content_feat).reshape(B, N, -1)
        styles = []
        for sf in styl_feat_list:
            assert sf.shape[0] == B and sf.shape[1] == N
            styles.append(sf if isinstance(sf, PackedCode) and not sf.stale and not self.use_double
                          else self._prep(sf).reshape(B, N, -1))
        one = lambda t, b: self.stats_code(t, b) if isinstance(t, PackedCode) else self.stats(t[b])      # noqa: E731
        stats = [(one(content_feat, b) if packed else self.stats(c[b]), [one(s, b) for s in styles]) for b in range(B)]
        affines, infos = [], []
        for cs, ss in stats:
            affines.append(self.factor(cs, ss, alpha_s_list, alpha_c, N))
            infos.append(self.last_info)
        if B > 1:
            # the reference factors the [B,N,N] stacks at once: if any sample needs jitter, every sample of the batch gets it
            # (cWCT.py:122-128).  Second factor call from the batch maximum of the retry counts (on the device: no host sync).
            need = torch.stack(infos).max(dim=0).values
            need[1] = 0
            affines = [self.factor(cs, ss, alpha_s_list, alpha_c, N, min_tries=need) for cs, ss in stats]
        if packed:
            return content_feat.with_affines(torch.stack(affines))
        out = torch.empty_like(c)
        for b in range(B):
            self.apply(c[b], affines[b], out=out[b])
        return out.to(in_dtype).reshape(B, N, cH, cW)

    # ------------------------------------------------------------------ cached-style extension
    def style_stats(self, style_feat):
        """Per-sample statistics of a style code [B,N,sH,sW] -> list of B stats tensors.  The reference
        re-encodes and re-factors the style for every frame (video_transfer.py:195); a video loop can
        compute this once per style and call transfer_with_stats per frame."""
        B, N = style_feat.shape[:2]
        packed = isinstance(style_feat, PackedCode) and not style_feat.stale and not self.use_double
        s = None if packed else self._prep(style_feat).reshape(B, N, -1)
        out = []
        for b in range(B):
            st = self.stats_code(style_feat, b) if packed else self.stats(s[b])
            if self.use_double:                      # (the prefactored record stores an fp32 factor)
                out.append(st)
                continue
            info = torch.zeros(1, dtype=torch.int32, device=st.device)
            with torch.cuda.device(st.device):       # Cholesky once per style, in place
                _lib.check(_lib.lib().vst_cwct_prefactor(_ptr(st), N, float(self.eps), _ptr(st), _ptr(info), _stream_ptr()),
                           "vst_cwct_prefactor")
            out.append(st)
        return out

    def transfer_with_stats(self, content_feat, style_stats, alpha_c=0.0, inplace=False):
        """transfer(content, style) with the style side given as style_stats(style) (len B or 1).  inplace=True
        overwrites a contiguous fp32 content code instead of allocating the result (like the reference's masked path,
        cWCT.py:62,103; one 128 MiB buffer less per 1024x1024 frame in flight)."""
        B, N, cH, cW = content_feat.shape
        if self._route_of(content_feat, masked=False) == "packed_rows":   # nothing is written here, the inverse pass applies the map
            affines = [self.factor(self.stats_code(content_feat, b), [style_stats[b if len(style_stats) > 1 else 0]], [1.0],
                                   alpha_c, N) for b in range(B)]
            return content_feat.with_affines(torch.stack(affines))
        in_dtype = content_feat.dtype
        c = self._prep(content_feat).reshape(B, N, -1)
        out = c if inplace and not isinstance(content_feat, PackedCode) and c.data_ptr() == content_feat.data_ptr() else torch.empty_like(c)
        for b in range(B):
            ss = style_stats[b if len(style_stats) > 1 else 0]
            affine = self.factor(self.stats(c[b]), [ss], [1.0], alpha_c, N)
            self.apply(c[b], affine, out=out[b])
        return out.to(in_dtype).reshape(B, N, cH, cW)

    def _transfer_seg(self, content_feat, style_feat, cmask, smask):
        """models/cWCT.py:49-109."""
        if self._route_of(content_feat, masked=True).startswith("masked_per_label"):
            # no matrix-core form at N = 16, no single-pass fp64 form: one statistics + apply pass per label (cWCT.py:83-103)
            return self._transfer_seg_per_label(content_feat, style_feat, cmask, smask)
        plan = self.plan_masks(cmask, smask, content_feat.shape, style_feat.shape, content_feat.device)
        return self.transfer_with_plan(content_feat, style_feat, plan)

    # ------------------------------------------------------------------ single-pass masked transfer
    # Everything `_transfer_seg` derives from the label maps (models/cWCT.py:72-76,166-189) happens on the device: both
    # histograms, the validity rule, and a label -> slot table (vst_label_plan); then ONE statistics pass per code for all
    # labels, one factor launch (a workgroup per label) and ONE apply pass.  No host histogram, no per-label launches, no
    # synchronisation: a per-frame mask costs its upload (or nothing, if it is already a device tensor).
    MAX_SLOTS = 32

    def _mask_to_device(self, m, hw, device, what):
        H, W = hw
        if torch.is_tensor(m):
            if m.numel() != H * W:
                raise ValueError(f"masks must have the feature resolution ({what} {tuple(m.shape)} vs {(H, W)})")
            return m.to(device=device, dtype=torch.uint8).reshape(-1).contiguous()
        m_np = np.asarray(m)
        if self.resize_masks and m_np.shape != (H, W):
            m_np = self.resize(np.ascontiguousarray(m_np.astype(np.uint8)), H, W)
        if m_np.size != H * W:
            raise ValueError(f"masks must have the feature resolution ({what} {m_np.shape} vs {(H, W)})")
        if m_np.dtype != np.uint8 and (m_np.max() > 255 or m_np.min() < 0):
            raise ValueError("labels must be in [0, 255]")
        return torch.from_numpy(np.ascontiguousarray(m_np.reshape(-1).astype(np.uint8))).to(device, non_blocking=True)

    def plan_masks(self, cmask, smask, content_shape, style_shape, device):
        """Device-side label plan per sample (numpy label maps as the reference hands them over, or uint8 device tensors).
        A video loop whose masks do not change builds this once; one whose masks change per frame pays two small
        histogram kernels per frame and no host work beyond the upload."""
        B, N, cH, cW = content_shape
        _, _, sH, sW = style_shape
        if N not in (32, 64, 128):
            raise NotImplementedError("the single-pass masked transfer needs N in (32, 64, 128)")
        L = _lib.lib()
        plan = MaskPlan()
        plan.shapes = (tuple(content_shape), tuple(style_shape))
        plan.max_slots = 0                # 0 = unknown (launches cover all 32 slots); learn_slots() tightens it
        plan.tables = []
        for b in range(B):
            cm = self._mask_to_device(cmask[b], (cH, cW), device, "content")
            sm = self._mask_to_device(smask[b], (sH, sW), device, "style")
            tab = torch.empty(2344, dtype=torch.uint8, device=device)
            with torch.cuda.device(device):
                _lib.check(L.vst_label_plan(_ptr(cm), cm.numel(), _ptr(sm), sm.numel(), _ptr(tab), _stream_ptr()), "vst_label_plan")
            plan.cm.append(cm)
            plan.sm.append(sm)
            plan.tables.append(tab)
        return plan

    @staticmethod
    def plan_info(plan, b=0):
        """(labels with a slot, overflow flag) of sample b — synchronises; for tests and diagnostics."""
        raw = plan.tables[b].cpu().numpy()
        n, over = int(raw[:4].view(np.int32)[0]), int(raw[4:8].view(np.int32)[0])
        return [int(v) for v in raw[8 + 2048 + 256: 8 + 2048 + 256 + n]], bool(over)

    def learn_slots(self, plan):
        """Read the slot counts back once (one synchronisation) so that later launches cover only the slots in use — worth
        it for a plan that is reused over a clip."""
        plan.max_slots = max(1, max(len(self.plan_info(plan, b)[0]) for b in range(len(plan.tables))))
        self._ensure_mask_rows(plan)
        return plan

    def _ensure_mask_rows(self, plan):
        """The content label maps in a PackedCode's row order, for plans the packed masked route can take (photorealistic
        codes, at most 8 slots).  Built HERE, once, and completed before returning: the plan is then shared by frames in
        flight on several streams (FramePipeline, bench.py), none of which may meet a half-written map."""
        B, N, cH, cW = plan.shapes[0]
        if plan.cm_rows is not None or N != 32 or not (1 <= int(plan.max_slots) <= 8):
            return
        L = _lib.lib()
        dev = plan.cm[0].device
        rows_all = []
        with torch.cuda.device(dev):
            for b in range(B):
                rows = torch.empty_like(plan.cm[b])
                _lib.check(L.vst_mask_to_code(_ptr(plan.cm[b]), _ptr(rows), cH, cW, _stream_ptr()), "vst_mask_to_code")
                rows_all.append(rows)
            torch.cuda.current_stream(dev).synchronize()
        plan.cm_rows = rows_all

    def _stats_labels(self, x2d, mask, table, max_slots):
        N, Lp = x2d.shape
        L = _lib.lib()
        out = torch.empty(self.MAX_SLOTS * (1 + N + N * N), dtype=torch.float64, device=x2d.device)
        ws = self._workspace(L.vst_cwct_labels_workspace_bytes(N, Lp), x2d.device)
        with torch.cuda.device(x2d.device):
            _lib.check(L.vst_cwct_stats_labels(_ptr(x2d), N, Lp, _ptr(mask), _ptr(table), int(max_slots), _ptr(out), _ptr(ws),
                                               _stream_ptr()), "vst_cwct_stats_labels")
        return out

    def bind_style(self, plan, style_feat):
        """Per-label statistics of the style code, computed once (the masked counterpart of style_stats): transfer_with_plan
        then skips the style side for every later frame.  Rebind when the style code changes."""
        B, N = style_feat.shape[:2]
        if tuple(style_feat.shape) != plan.shapes[1]:
            raise ValueError(f"plan was made for a style code of shape {plan.shapes[1]}, got {tuple(style_feat.shape)}")
        s = self._prep(style_feat).reshape(B, N, -1)
        plan.style = [self._stats_labels(s[b], plan.sm[b], plan.tables[b], plan.max_slots) for b in range(B)]
        return plan

    def transfer_with_plan(self, content_feat, style_feat, plan, inplace=False):
        """transfer(content, style, cmask, smask) with the mask work given as plan_masks(...) (and, after bind_style,
        the style side too; style_feat may then be None).  Pixels whose label has no slot keep the content feature."""
        B, N, cH, cW = content_feat.shape
        if tuple(content_feat.shape) != plan.shapes[0]:
            raise ValueError(f"plan was made for a content code of shape {plan.shapes[0]}, got {tuple(content_feat.shape)}")
        if self.use_double:
            raise NotImplementedError("transfer_with_plan (this repo's cached-mask extension) has no fp64 form: with "
                                      "use_double=True call transfer(content, style, cmask, smask)")
        if self._route_of(content_feat, masked=True, max_slots=plan.max_slots) == "masked_packed_rows":
            return self._transfer_with_plan_packed(content_feat, style_feat, plan)
        if self.last_route != "masked_single_pass":
            raise NotImplementedError("transfer_with_plan needs N in (32, 64, 128)")
        in_dtype = content_feat.dtype
        c = self._prep(content_feat).reshape(B, N, -1)
        s = None
        if plan.style is None:
            if style_feat is None or tuple(style_feat.shape) != plan.shapes[1]:
                raise ValueError("transfer_with_plan needs the style code the plan was made for (or bind_style first)")
            s = self._prep(style_feat).reshape(B, N, -1)
        out = c if inplace and not isinstance(content_feat, PackedCode) and c.data_ptr() == content_feat.data_ptr() else torch.empty_like(c)
        L = _lib.lib()
        ms = int(plan.max_slots)
        for b in range(B):
            tab = plan.tables[b]
            cs = self._stats_labels(c[b], plan.cm[b], tab, ms)
            ss = plan.style[b] if plan.style is not None else self._stats_labels(s[b], plan.sm[b], tab, ms)
            affines = torch.empty(self.MAX_SLOTS * (N * N + N), dtype=torch.float32, device=c.device)
            info = torch.empty(self.MAX_SLOTS * 3, dtype=torch.int32, device=c.device)
            with torch.cuda.device(c.device):
                _lib.check(L.vst_cwct_factor_labels(_ptr(cs), _ptr(ss), _ptr(tab), ms, float(self.eps), N, _ptr(affines),
                                                    _ptr(info), _stream_ptr()), "vst_cwct_factor_labels")
                prec = {"fp32": _lib.PREC_FP32, "bf16x3": _lib.PREC_BF16X3, "f16x2": _lib.PREC_F16X2, "f16x2h": _lib.PREC_F16X2H}[self.precision]
                _lib.check(L.vst_cwct_apply_labels(_ptr(c[b]), _ptr(out[b]), N, c.shape[2], _ptr(affines), _ptr(plan.cm[b]),
                                                   _ptr(tab), ms, prec, _stream_ptr()), "vst_cwct_apply_labels")
            self.last_info = info
        return out.to(in_dtype).reshape(B, N, cH, cW)

    def _transfer_with_plan_packed(self, content, style_feat, plan):
        """transfer_with_plan on a PackedCode (photorealistic codes, at most 8 label slots, known after learn_slots): the
        per-label statistics run on the packed rows with the label map in the rows' order (made once per plan), and the result
        is the same rows with the per-row maps pending - the inverse pass applies them while it loads its state."""
        B, N, cH, cW = content.shape
        L = _lib.lib()
        ms = int(plan.max_slots)
        dev = content.packed.device
        self._ensure_mask_rows(plan)          # (a plan whose max_slots was set by hand: built and completed now)
        s = None
        if plan.style is None:
            if style_feat is None or tuple(style_feat.shape) != plan.shapes[1]:
                raise ValueError("transfer_with_plan needs the style code the plan was made for (or bind_style first)")
            s = self._prep(style_feat).reshape(B, N, -1)
        per_image = []
        ws = self._workspace(L.vst_cwct_stats_labels_code_workspace_bytes(cH, cW), dev)
        for b in range(B):
            tab = plan.tables[b]
            cs = torch.empty(self.MAX_SLOTS * (1 + N + N * N), dtype=torch.float64, device=dev)
            ss = plan.style[b] if plan.style is not None else self._stats_labels(s[b], plan.sm[b], tab, ms)
            affines = torch.empty(self.MAX_SLOTS * (N * N + N), dtype=torch.float32, device=dev)
            info = torch.empty(self.MAX_SLOTS * 3, dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(L.vst_cwct_stats_labels_code(_ptr(content.packed[b]), cH, cW, _ptr(plan.cm_rows[b]), _ptr(tab), ms,
                                                        _ptr(cs), _ptr(ws), _stream_ptr()), "vst_cwct_stats_labels_code")
                _lib.check(L.vst_cwct_factor_labels(_ptr(cs), _ptr(ss), _ptr(tab), ms, float(self.eps), N, _ptr(affines),
                                                    _ptr(info), _stream_ptr()), "vst_cwct_factor_labels")
            self.last_info = info
            per_image.append((affines, plan.cm_rows[b], tab))
        return content.with_label_affines(per_image, ms)

    # ------------------------------------------------------------------ per-label form (N = 16 only)
    def _transfer_seg_per_label(self, content_feat, style_feat, cmask, smask):
        B, N, cH, cW = content_feat.shape
        _, _, sH, sW = style_feat.shape
        in_dtype = content_feat.dtype
        c = self._prep(content_feat).reshape(B, N, -1)
        s = self._prep(style_feat).reshape(B, N, -1)
        out = c.clone()
        for b in range(B):
            cm_np, sm_np = np.asarray(cmask[b]), np.asarray(smask[b])
            if cm_np.size != cH * cW or sm_np.size != sH * sW:
                raise ValueError("masks must have the feature resolution")
            label_set, label_indicator = self.compute_label_info(cm_np, sm_np)
            cm = torch.from_numpy(np.ascontiguousarray(cm_np.reshape(-1).astype(np.uint8))).to(c.device)
            sm = torch.from_numpy(np.ascontiguousarray(sm_np.reshape(-1).astype(np.uint8))).to(c.device)
            for label in label_set:
                if not label_indicator[label]:
                    continue
                affine = self.factor(self.stats(c[b], cm, int(label)), [self.stats(s[b], sm, int(label))], [1.0], 0.0, N)
                self.apply(c[b], affine, out=out[b], mask=cm, label=int(label))
        return out.to(in_dtype).reshape(B, N, cH, cW)

    # ------------------------------------------------------------------ helpers (public by convention)
    def cholesky_dec(self, conv, invert=False):
        """models/cWCT.py:111-132 for one [N,N] matrix (N in {16,32,64,128})."""
        N = conv.shape[-1]
        dev = conv.device
        st = torch.zeros(1 + N + N * N, dtype=torch.float64, device=dev)
        st[0] = 2.0
        st[1 + N:] = conv.detach().double().reshape(-1)
        ident = self._identity_stats(N, dev)
        if invert:      # T = I * L^-1
            aff = self.factor(st, [ident], [1.0], 0.0, N)
        else:           # T = L * I^-1
            aff = self.factor(ident, [st], [1.0], 0.0, N)
        return aff[: N * N].reshape(N, N).to(conv.dtype)

    def whitening(self, x):
        """models/cWCT.py:134-149 on a 2-D [N,L] matrix."""
        x = self._prep(x)
        N = x.shape[0]
        aff = self.factor(self.stats(x), [self._identity_stats(N, x.device)], [1.0], 0.0, N)
        return self.apply(x, aff)

    def coloring(self, content_whiten_feat, style_feat):
        """models/cWCT.py:152-164 on 2-D matrices."""
        w = self._prep(content_whiten_feat)
        s = self._prep(style_feat)
        N = w.shape[0]
        aff = self.factor(self._identity_stats(N, w.device), [self.stats(s)], [1.0], 0.0, N)
        return self.apply(w, aff)

    def compute_label_info(self, content_seg, style_seg):
        """models/cWCT.py:166-189 (histograms instead of one np.where per label; same result)."""
        content_seg, style_seg = np.asarray(content_seg), np.asarray(style_seg)
        if content_seg.size == 0 or style_seg.size == 0:
            return
        max_label = int(np.max(content_seg)) + 1
        ch = np.bincount(content_seg.reshape(-1).astype(np.int64), minlength=max_label)
        sh = np.bincount(style_seg.reshape(-1).astype(np.int64), minlength=max_label)
        label_set = np.nonzero(ch)[0].astype(content_seg.dtype)
        label_indicator = np.zeros(max_label)
        for l in label_set:
            a, b = int(ch[l]), int(sh[l])
            label_indicator[l] = a > 10 and b > 10 and a / b < 100 and b / a < 100
        return label_set, label_indicator

    def resize(self, img, H, W):
        """models/cWCT.py:191-197 (NEAREST)."""
        from PIL import Image
        if len(img.shape) == 2:
            return np.array(Image.fromarray(img).resize((W, H), Image.NEAREST))
        return np.array(Image.fromarray(img, mode='RGB').resize((W, H), Image.NEAREST))

    def get_index(self, feat, label):
        """models/cWCT.py:199-204."""
        mask = np.where(feat.reshape(feat.shape[0] * feat.shape[1]) == label)
        if mask[0].size <= 0:
            return None
        return torch.LongTensor(mask[0])
