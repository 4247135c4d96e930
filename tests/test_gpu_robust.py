"""GPU robustness tests of the conv arithmetic modes against the oracle (VERDICT r2 "next round" item 1).

The regular parity tests run on friendly synthetic weights (O(1), well-conditioned codes).  Here the same oracle comparison
runs on hostile checkpoints (tests/robust.py) and inputs, in every mode a user can select:

  (a) an ILL-CONDITIONED code: per-channel std of z spanning >= 1e3, cond(cov z) >= 1e5.  Budget (BASELINE.json north_star):
      rel-L2 <= 1e-3 on z_c, z_cs and the stylised frame, and the worst single channel of z_cs within 1e-2 of its own norm —
      the quantity whitening (models/cWCT.py:134-149) amplifies and a per-tensor norm hides.
  (b) scales: inputs x 1e3 and x 1/16; intermediates rescaled by 1e3 / 1e-3 (function-preserving, see tests/robust.py).
  (c) the device-side fp16 range flag: saturation is DETECTED (vst_range_flags, RevResNet.check_range, the load-time
      calibration), never silent.

Outcome that the asserts below pin: `bf16x3` (the product default) holds the budget everywhere.  `f16x2` / `f16x2h` hold it on
well-conditioned codes and under every rescaling (the pack-time exponent normalisation makes them scale-free), but NOT on the
ill-conditioned code: their 11-bit weights are a 2^-12 relative perturbation of the model, which that checkpoint amplifies to
1.5e-3 .. 4e-3 on the stylised frame (the fp32 oracle itself is 5e-5 from fp64 there).  That is why they are opt-in.
Each case appends its measured errors to gpurun_out/robustness.jsonl (copied to profiles/ by the builder).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref
from vstnet_amd import _lib
from vstnet_amd.synth import synthetic_state_dict, synthetic_frames
from tests.robust import (ramp_state_dict, rescale_state_dict, code_conditioning, rel_l2, max_rel, worst_channel)

pytestmark = pytest.mark.gpu
BUDGET = 1e-3                 # north_star: within 1e-3 relative of the reference
BUDGET_CHANNEL = 1e-2         # worst channel of z_cs
MODES = ["bf16x3", "f16x2", "f16x2h"]
# whole-net bounds on the friendly checkpoint (tests/test_gpu_parity.py NET_TOL); what the scale cases must still hold
NOMINAL = {"bf16x3": 5e-5, "f16x2": 2.5e-4, "f16x2h": 3.5e-4}
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _net(sd, mode, precision, **kw):
    from models.RevResNet import RevResNet
    hd, sp = (16, 2) if mode == "photo" else (64, 1)
    net = RevResNet(hidden_dim=hd, sp_steps=sp, precision=precision)
    for k, v in kw.items():
        setattr(net, k, v)
    net.load_state_dict(sd)
    return net.to("cuda").eval(), sp


def _record(case, **vals):
    d = os.path.join(REPO, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "robustness.jsonl"), "a") as f:
            f.write(json.dumps({"case": case, **{k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in vals.items()}}) + "\n")
    except OSError:
        pass


def _stylize_gpu(net, xc, xs):
    from models.cWCT import cWCT
    cw = cWCT(precision=net.precision)
    with torch.no_grad():
        zc, zs = net(xc.cuda()), net(xs.cuda())
        zcs = cw.transfer(zc, zs)
        sty = net(zcs, forward=False)
        return [t.materialize() if hasattr(t, "materialize") else t for t in (zc, zs, zcs)] + [sty]


_ORACLE = {}


def _oracle(key, xc, xs, sd, sp):
    if key not in _ORACLE:
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
        with torch.no_grad():
            _ORACLE[key] = cpu_ref.stylize(xc, xs, sd, sp)
    return _ORACLE[key]


def test_default_precision_is_the_fp32_class_mode():
    from models.RevResNet import RevResNet
    from models.cWCT import cWCT
    if "VST_PRECISION" not in os.environ:
        assert _lib.default_precision() == "bf16x3" and RevResNet().precision == "bf16x3" and cWCT().precision == "bf16x3"


# ------------------------------------------------------------------------------------------- (a) ill-conditioned code
# bounds per (mode, span): (rel-L2 of z_c / z_cs / stylised, worst channel of z_cs).  The default mode must hold the BUDGET;
# the fp16 modes' bounds at span 1e3 are what they measure plus margin — beyond the budget, which keeps them opt-in.
ILL_BOUNDS = {
    ("bf16x3", 10.0): (5e-5, 5e-5, 5e-5, 2e-4), ("bf16x3", 1e3): (1e-4, 1e-4, BUDGET, 1e-3),
    ("f16x2", 10.0): (5e-4, 5e-4, 2e-4, 2e-3), ("f16x2", 1e3): (BUDGET, BUDGET, 5e-3, BUDGET_CHANNEL),
    ("f16x2h", 10.0): (6e-4, 6e-4, 4e-4, 3e-3), ("f16x2h", 1e3): (1.5e-3, 1.5e-3, 3e-2, BUDGET_CHANNEL),
}


@pytest.mark.parametrize("span", [10.0, 1e3])
@pytest.mark.parametrize("precision", MODES)
@pytest.mark.parametrize("mode", ["photo", "art"])
def test_ill_conditioned_code(mode, precision, span):
    hd, sp, n_code = (16, 2, 32) if mode == "photo" else (64, 1, 128)
    sd = ramp_state_dict(synthetic_state_dict(1234, hd, sp), span, n_code)
    H = W = 96
    xc, xs = synthetic_frames(1, H, W, seed=0), synthetic_frames(1, H, W, seed=1)
    ref = _oracle(("ill", mode, span), xc, xs, sd, sp)
    smin, smax, cond = code_conditioning(ref[0])
    if span >= 1e3:
        assert smax / smin >= 5e2 and cond >= 1e5, (smin, smax, cond)     # the case really is ill-conditioned
    net, _ = _net(sd, mode, precision)
    got = _stylize_gpu(net, xc, xs)
    e_zc, e_zcs, e_sty = rel_l2(got[0], ref[0]), rel_l2(got[2], ref[2]), rel_l2(got[3], ref[3])
    e_ch = worst_channel(got[2], ref[2])
    _record("ill_conditioned", mode=mode, precision=precision, span=span, std_min=smin, std_max=smax, cond=cond,
            zc=e_zc, zcs=e_zcs, stylized=e_sty, stylized_max_rel=max_rel(got[3], ref[3]), zcs_worst_channel=e_ch,
            within_budget=bool(max(e_zc, e_zcs, e_sty) <= BUDGET and e_ch <= BUDGET_CHANNEL))
    b = ILL_BOUNDS[(precision, span)]
    assert e_zc <= b[0] and e_zcs <= b[1] and e_sty <= b[2] and e_ch <= b[3], \
        f"{mode}/{precision} span {span:g}: z_c {e_zc:.2e} z_cs {e_zcs:.2e} stylised {e_sty:.2e} worst channel {e_ch:.2e} (bounds {b})"
    if precision == _lib.default_precision():
        # the rule of VERDICT r2 item 1: the DEFAULT mode holds the north_star budget on the ill-conditioned code
        assert max(e_zc, e_zcs, e_sty) <= BUDGET and e_ch <= BUDGET_CHANNEL


@pytest.mark.parametrize("span,expect", [(None, ("f16x2h",)), (10.0, ("f16x2h", "f16x2")), (1e3, ("bf16x3",))])
def test_auto_precision_picks_a_mode_that_holds_the_budget(span, expect):
    """precision='auto': at (re)pack time a probe stylisation per candidate mode is compared ON THE DEVICE with bf16x3; the
    fastest mode within a quarter of the budget is taken.  On the friendly checkpoint that is f16x2h, on the ill-conditioned
    one bf16x3 — and whatever it picks must hold the 1e-3 budget against the ORACLE on frames other than the probe."""
    sd = synthetic_state_dict(1234, 16, 2)
    if span is not None:
        sd = ramp_state_dict(sd, span, 32)
    net, _ = _net(sd, "photo", "auto")
    xc, xs = synthetic_frames(1, 96, 96, seed=0), synthetic_frames(1, 96, 96, seed=1)
    ref = _oracle(("ill", "photo", span) if span is not None else ("friendly",), xc, xs, sd, 2)
    got = _stylize_gpu(net, xc, xs)
    assert net.resolved_precision in expect, (net.resolved_precision, net.calibration)
    errs = [rel_l2(got[i], ref[i]) for i in (0, 2, 3)]
    _record("auto_precision", span=span, chosen=net.resolved_precision, zc=errs[0], zcs=errs[1], stylized=errs[2],
            calibration=net.calibration)
    assert max(errs) <= BUDGET and worst_channel(got[2], ref[2]) <= BUDGET_CHANNEL, (net.resolved_precision, errs)
    rec = net.calibrate(xc, xs)                                   # the caller's own frames: same verdict here
    assert rec["chosen"] in expect and net.resolved_precision == rec["chosen"]
    # a repack (new weights) decides again
    net.load_state_dict(ramp_state_dict(synthetic_state_dict(1234, 16, 2), 1e3, 32))
    net(xc.cuda())
    assert net.resolved_precision == "bf16x3"
    fixed, _ = _net(sd, "photo", "bf16x3")
    with pytest.raises(RuntimeError):
        fixed.calibrate()


# ------------------------------------------------------------------------------------------- (b) scales
@pytest.mark.parametrize("f1,f2", [(1e3, 1e-3), (1e-3, 1e3), (2.0 ** 12, 2.0 ** 12)])
@pytest.mark.parametrize("precision", MODES)
def test_rescaled_intermediates(precision, f1, f2):
    """h1 x f1, h2 x f2, compensated in the next conv: the same function.  The pack-time exponent normalisation
    (vst_normalize_block) undoes any such rescaling, so every mode holds its nominal bound and nothing saturates."""
    sd0 = synthetic_state_dict(1234, 16, 2)
    sd = rescale_state_dict(sd0, f1, f2)
    xc, xs = synthetic_frames(1, 96, 96, seed=0), synthetic_frames(1, 96, 96, seed=1)
    ref = _oracle(("rescale", f1, f2), xc, xs, sd, 2)
    net, _ = _net(sd, "photo", precision)
    _lib.range_flags(reset=True)
    got = _stylize_gpu(net, xc, xs)
    flags = _lib.range_flags(reset=True)
    errs = [rel_l2(got[i], ref[i]) for i in (0, 2, 3)]
    _record("rescaled_intermediates", precision=precision, f1=f1, f2=f2, zc=errs[0], zcs=errs[1], stylized=errs[2], flags=flags)
    assert max(errs) <= NOMINAL[precision], (precision, f1, f2, errs)
    assert flags == 0


@pytest.mark.parametrize("precision", ["f16x2", "f16x2h"])
def test_rescaled_intermediates_without_normalisation_is_why(precision):
    """the same checkpoint with the normalisation switched off: the fp16 operand paths lose the low part of h2 (x 1e-3: the
    lo plane underflows) — measurably outside the budget, or flagged.  Documents what vst_normalize_block is for."""
    sd = rescale_state_dict(synthetic_state_dict(1234, 16, 2), 1e3, 1e-3)
    xc, xs = synthetic_frames(1, 96, 96, seed=0), synthetic_frames(1, 96, 96, seed=1)
    ref = _oracle(("rescale", 1e3, 1e-3), xc, xs, sd, 2)
    net, _ = _net(sd, "photo", precision, normalize_intermediates=False, calibrate_on_load=False)
    _lib.range_flags(reset=True)
    got = _stylize_gpu(net, xc, xs)
    flags = _lib.range_flags(reset=True)
    e = rel_l2(got[0], ref[0])
    _record("rescaled_no_normalisation", precision=precision, zc=e, flags=flags)
    assert e > BUDGET or flags != 0


@pytest.mark.parametrize("scale", [1e3, 1.0 / 16])
@pytest.mark.parametrize("precision", MODES)
def test_input_scales(precision, scale):
    """content and style x 1e3 (state ~1e3: inside fp16's range, above the lo plane's underflow) and x 1/16 (a dark frame: the
    stylised frame's own scale drops 16x while the bias-driven part of the network's error does not, hence twice the nominal
    bound on that tensor — measured 4.4e-4 for f16x2h, 2.2e-5 for bf16x3; the codes stay at their nominal error)"""
    sd = synthetic_state_dict(1234, 16, 2)
    xc, xs = synthetic_frames(1, 96, 96, seed=0) * scale, synthetic_frames(1, 96, 96, seed=1) * scale
    ref = _oracle(("inscale", scale), xc, xs, sd, 2)
    net, _ = _net(sd, "photo", precision)
    _lib.range_flags(reset=True)
    got = _stylize_gpu(net, xc, xs)
    flags = _lib.range_flags(reset=True)
    errs = [rel_l2(got[i], ref[i]) for i in (0, 2, 3)]
    _record("input_scale", precision=precision, scale=scale, zc=errs[0], zcs=errs[1], stylized=errs[2], flags=flags)
    assert max(errs[:2]) <= NOMINAL[precision] and errs[2] <= NOMINAL[precision] * (2 if scale < 1 else 1), (precision, scale, errs)
    assert max(errs) <= BUDGET
    assert flags == 0


# ------------------------------------------------------------------------------------------- (c) the range flag
def test_saturation_is_flagged_not_silent(tmp_path):
    """inputs x 3e5 push the state beyond +-65504: the fp16 modes clamp AND raise VST_RANGE_SATURATED (query API,
    RevResNet.check_range); bf16x3 has no such range and stays exact; the flag is sticky until reset."""
    sd = synthetic_state_dict(1234, 16, 2)
    x = synthetic_frames(1, 64, 64, seed=3) * 3e5
    with torch.no_grad():
        zr = cpu_ref.revnet_forward(x, sd, 2)
    assert float(zr.abs().max()) > 65504
    for precision in ("f16x2", "f16x2h"):
        net, _ = _net(sd, "photo", precision)
        assert net.check_range() == 0                                      # the [0,1) probe is fine
        assert net.check_range(x) & _lib.RANGE_SATURATED                   # this input is not
        _lib.range_flags(reset=True)
        net(x.cuda())
        assert _lib.range_flags() & _lib.RANGE_SATURATED and _lib.range_flags(reset=True) & _lib.RANGE_SATURATED
        assert _lib.range_flags() == 0                                      # cleared by the reset
    net, _ = _net(sd, "photo", "bf16x3")
    _lib.range_flags(reset=True)
    z = net(x.cuda())
    assert _lib.range_flags() == 0
    assert rel_l2(z, zr) <= 5e-5


def test_frame_pipeline_reports_saturation_per_frame():
    """the video loop (vstnet_amd/pipeline.py) carries the range flags with every frame: a clip that saturates an fp16 operand
    fails at the frame that did it (here: a checkpoint with h1 scaled by 2^18 and the pack-time normalisation and the load-time
    probe both switched off, i.e. exactly the case those two exist to prevent); bf16x3 runs the same clip"""
    from models.cWCT import cWCT
    from vstnet_amd.pipeline import FramePipeline
    sd = rescale_state_dict(synthetic_state_dict(1234, 16, 2), 2.0 ** 18, 1.0)
    frames = [(synthetic_frames(1, 32, 48, seed=60 + i)[0].permute(1, 2, 0) * 255).byte().numpy() for i in range(4)]
    for precision in ("f16x2h", "bf16x3"):
        net, _ = _net(sd, "photo", precision, normalize_intermediates=False, calibrate_on_load=False)
        cw = cWCT(precision=precision)
        _lib.range_flags(reset=True)
        with torch.no_grad():
            stats = cw.style_stats(net.forward_u8(torch.from_numpy(frames[0])[None].cuda()))
        _lib.range_flags(reset=True)
        pipe = FramePipeline(net, lambda z, i: cw.transfer_with_stats(z, stats), 32, 48, depth=2, compute_streams=1)
        got = []
        if precision == "bf16x3":
            assert pipe.run(frames, lambda i, a: got.append(i)) == 4 and got == [0, 1, 2, 3]
        else:
            with pytest.raises(RuntimeError, match="range flags"):
                pipe.run(frames, lambda i, a: got.append(i))
            assert got == []                                   # the first frame already saturated: nothing was handed on
    _lib.range_flags(reset=True)


def test_load_time_calibration_rejects_an_unrepresentable_checkpoint():
    """a conv weight of 1e6 (fp16: Inf) with the normalisation off: the fp16 modes refuse the checkpoint at the first call
    (RuntimeError naming bf16x3), bf16x3 runs it; with the normalisation on the same checkpoint is fine in every mode."""
    sd = synthetic_state_dict(1234, 16, 2)
    sd = rescale_state_dict(sd, 2.0 ** 24, 1.0)         # W1 rows ~ 1e6, compensated in W4: the same function
    x = synthetic_frames(1, 32, 32, seed=1)
    with torch.no_grad():
        zr = cpu_ref.revnet_forward(x, sd, 2)
    for precision in ("f16x2", "f16x2h"):
        net, _ = _net(sd, "photo", precision, normalize_intermediates=False)
        with pytest.raises(RuntimeError, match="bf16x3"):
            net(x.cuda())
        net, _ = _net(sd, "photo", precision)
        assert rel_l2(net(x.cuda()), zr) <= NOMINAL[precision]
    net, _ = _net(sd, "photo", "bf16x3", normalize_intermediates=False)
    assert rel_l2(net(x.cuda()), zr) <= 5e-5
    assert _lib.range_flags(reset=True) & _lib.RANGE_SATURATED == 0


def test_normalisation_is_exact_in_the_fp32_class_modes():
    """vst_normalize_block scales by powers of two: bf16x3 / fp32 results are bit-identical with and without it"""
    sd = synthetic_state_dict(1234, 16, 2)
    x = synthetic_frames(2, 40, 56, seed=5).cuda()
    for precision in ("bf16x3", "fp32"):
        a, _ = _net(sd, "photo", precision)
        b, _ = _net(sd, "photo", precision, normalize_intermediates=False)
        za, zb = a(x), b(x)
        assert torch.equal(za, zb), precision
        assert torch.equal(a(za, forward=False), b(zb, forward=False))


def test_normalize_block_scales(tmp_path):
    """the C entry point itself: scales are powers of two from the rows' L2 norms, the block's function is unchanged"""
    import ctypes as C
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    cin, mid, cout = 64, 16, 64
    w1 = torch.randn(mid, cin, 3, 3, generator=g) * torch.logspace(-3, 3, mid)[:, None, None, None]
    b1, w4 = torch.randn(mid, generator=g), torch.randn(mid, mid, 3, 3, generator=g) * 0.01
    b4, w7 = torch.randn(mid, generator=g), torch.randn(cout, mid, 3, 3, generator=g)
    w1[3] = 0                                                              # a dead channel: scale 1
    dev = [t.clone().cuda() for t in (w1, b1, w4, b4, w7)]
    sc = torch.zeros(2 * mid, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(L.vst_normalize_block(p(dev[0]), p(dev[1]), p(dev[2]), p(dev[3]), p(dev[4]), cin, mid, cout, p(sc),
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vst_normalize_block")
    s = sc.cpu()
    assert torch.all(torch.log2(s) == torch.round(torch.log2(s))) and float(s[3]) == 1.0
    n1 = dev[0].cpu().flatten(1).norm(dim=1)
    n4 = dev[2].cpu().flatten(1).norm(dim=1)
    live = torch.arange(mid) != 3
    assert torch.all((n1[live] > 0.70) & (n1[live] < 1.42)) and torch.all((n4 > 0.70) & (n4 < 1.42))
    sd_a = {"conv.1.weight": w1, "conv.1.bias": b1, "conv.4.weight": w4, "conv.4.bias": b4, "conv.7.weight": w7,
            "conv.7.bias": torch.zeros(cout)}
    sd_b = dict(zip(("conv.1.weight", "conv.1.bias", "conv.4.weight", "conv.4.bias", "conv.7.weight"), [t.cpu() for t in dev]))
    sd_b["conv.7.bias"] = torch.zeros(cout)
    x = torch.randn(1, cin, 12, 12, generator=g, dtype=torch.float64)
    fa = cpu_ref.residual_F(x, {k: v.double() for k, v in sd_a.items()}, "", 1)
    fb = cpu_ref.residual_F(x, {k: v.double() for k, v in sd_b.items()}, "", 1)
    assert rel_l2(fb, fa) < 1e-12
    assert L.vst_normalize_block(None, None, None, None, None, 4, 4, 4, None, None) == -1
    assert L.vst_normalize_block(p(dev[0]), p(dev[1]), p(dev[2]), p(dev[3]), p(dev[4]), cin, 65, cout, None, None) == -2
