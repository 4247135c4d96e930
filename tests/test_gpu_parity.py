"""GPU parity tests: the HIP path (through the C ABI of libvstnet_hip.so) against the oracle
(oracle/cpu_ref.py) and the golden vectors minted from the reference.

Tolerance (BASELINE.json north_star): outputs within 1e-3 relative (fp32) of the reference, checked
both as rel-L2 and as max|d|/max|ref|.  The bf16x3 split-precision convs land near 3e-6, so the
asserts below use tighter budgets where the case allows, to catch regressions early.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref
from vstnet_amd import _lib
from vstnet_amd.synth import synthetic_state_dict, synthetic_frames, synthetic_mask
from tests.zc import nchw_to_zc, zc_to_nchw, ptr, stream, rel_err

pytestmark = pytest.mark.gpu
T = torch.from_numpy
TOL = 1e-3            # the north_star budget
TIGHT = 5e-5          # what the bf16x3 path is expected to hold on these weights


@pytest.fixture(scope="module")
def L():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _lib.lib()


def make_net(mode="photo", precision=None, seed=1234):
    """precision=None: the PRODUCT default (vstnet_amd._lib.default_precision), i.e. what a user of the drop-in classes runs"""
    from models.RevResNet import RevResNet
    hd, sp = (16, 2) if mode == "photo" else (64, 1)
    net = RevResNet(hidden_dim=hd, sp_steps=sp, precision=precision)
    sd = synthetic_state_dict(seed, hd, sp)
    net.load_state_dict(sd)
    return net.to("cuda").eval(), sd, sp


def assert_close(got, ref, tol, what, tol_max=None):
    l2, mx = rel_err(got, ref)
    tol_max = tol if tol_max is None else tol_max
    assert l2 <= tol and mx <= tol_max, f"{what}: rel-L2 {l2:.3e}, max-rel {mx:.3e} (tol {tol:g}/{tol_max:g})"
    return l2, mx


# ------------------------------------------------------------------------------------------- glue
@pytest.mark.parametrize("shape", [(1, 3, 8, 8), (2, 3, 24, 40), (1, 3, 64, 68), (1, 5, 16, 132)])
def test_pack_unpack_input(L, shape):
    B, Cc, H, W = shape
    x = torch.rand(shape, device="cuda")
    s1 = torch.full((B, H // 4, W // 4, 256), 7.0, device="cuda")
    s2 = torch.full_like(s1, 7.0)
    _lib.check(L.vst_pack_input(ptr(x), ptr(s1), ptr(s2), B, Cc, H, W, stream()), "pack")
    ref = cpu_ref.inj_pad_fwd(x.cpu(), 32 - Cc)
    assert torch.equal(zc_to_nchw(s1.cpu(), 16), ref[:, :16])
    assert torch.equal(zc_to_nchw(s2.cpu(), 16), ref[:, 16:])
    y = torch.empty_like(x)
    _lib.check(L.vst_unpack_output(ptr(s1), ptr(y), B, Cc, H, W, stream()), "unpack")
    assert torch.equal(y, x)


@pytest.mark.parametrize("sp", [1, 2])
@pytest.mark.parametrize("shape", [(1, 8, 8), (2, 24, 40), (1, 64, 72), (1, 16, 264)])
def test_spread_gather(L, sp, shape):
    B, H, W = shape
    s1 = torch.randn(B, H // 4, W // 4, 256, device="cuda")
    s2 = torch.randn_like(s1)
    zshape = (B, 32, H, W) if sp == 2 else (B, 128, H // 2, W // 2)
    z = torch.full(zshape, -9.0, device="cuda")
    _lib.check(L.vst_spread(ptr(s1), ptr(s2), ptr(z), B, H, W, sp, stream()), "spread")
    m = cpu_ref.merge(zc_to_nchw(s1.cpu(), 256), zc_to_nchw(s2.cpu(), 256))
    ref = m
    for _ in range(sp):
        ref = cpu_ref.unsqueeze(ref)
    assert torch.equal(z.cpu(), ref)
    g1, g2 = torch.zeros_like(s1), torch.zeros_like(s2)
    _lib.check(L.vst_gather(ptr(z), ptr(g1), ptr(g2), B, H, W, sp, stream()), "gather")
    assert torch.equal(g1, s1) and torch.equal(g2, s2)


# ------------------------------------------------------------------------------------------- blocks
BLOCKS = [("c16s1", 3, 16, 1), ("c64s2", 10, 64, 2), ("c64s1", 13, 64, 1), ("c256s2", 20, 256, 2),
          ("c256s1", 25, 256, 1), ("cr0", 30, 256, 1)]


def run_block(L, net, k, channel, stride, direction, precision, dst_nchw, src_nchw, H, W):
    """dst/src: NCHW views at their own levels -> returns updated dst view (same level as given)."""
    B = dst_nchw.shape[0]
    dst = nchw_to_zc(dst_nchw).cuda()
    src = nchw_to_zc(src_nchw).cuda()
    tmp = torch.empty(L.vst_block_tmp_bytes(B, H, W), dtype=torch.uint8, device="cuda")
    w = net._ensure_packed(torch.device("cuda", torch.cuda.current_device()))
    rc = L.vst_block_apply(C.byref(w.blocks[k]), channel, stride, direction, precision, ptr(dst), ptr(src), ptr(tmp),
                           B, H, W, stream())
    _lib.check(rc, "vst_block_apply")
    return zc_to_nchw(dst.cpu(), dst_nchw.shape[1])


# F16X2 rounds every block's weights to fp16 (2^-12): ~1e-4 of F per block on these fixtures; F16X2H also takes the conv inputs
# that cross HBM as fp16.  Budgets here: 3e-4 / 4e-4 of the block's output.
@pytest.mark.parametrize("precision,tol", [(_lib.PREC_FP32, 2e-6), (_lib.PREC_BF16X3, 2e-5), (_lib.PREC_F16X2, 3e-4),
                                           (_lib.PREC_F16X2H, 4e-4)])
@pytest.mark.parametrize("name,k,channel,stride", BLOCKS)
def test_block_golden(L, golden, name, k, channel, stride, precision, tol):
    g = golden("blocks")
    net, sd, _ = make_net("photo")
    x1, x2 = T(g[f"{name}_x1"]), T(g[f"{name}_x2"])
    lv = {16: 0, 64: 1, 256: 2}[channel]
    H, W = x1.shape[2] << (lv - (stride == 2)), x1.shape[3] << (lv - (stride == 2))
    # forward: y1 = F(x2) + squeeze?(x1); dst holds x1 (finer view for stride 2 — same memory)
    y1 = run_block(L, net, k, channel, stride, +1, precision, x1, x2, H, W)
    ref_y1 = T(g[f"{name}_out_y1"])
    if stride == 2:
        y1 = cpu_ref.squeeze(y1)
    assert_close(y1, ref_y1, tol, f"{name} forward y1")
    # inverse: x1 = y1 - F(x2)
    src = T(g[f"{name}_out_x2"])
    if stride == 2:
        src = cpu_ref.unsqueeze(src)
    x1r = run_block(L, net, k, channel, stride, -1, precision, ref_y1, src, H, W)
    ref_x1 = T(g[f"{name}_inv_x1"])
    if stride == 2:
        x1r = cpu_ref.unsqueeze(x1r)
    assert_close(x1r, ref_x1, tol, f"{name} inverse x1")


# ------------------------------------------------------------------------------------------- network
# f16x2: weights rounded to fp16 once (measured 1.44e-4 rel-L2 / 1.51e-4 max-rel on the 1024x1024 code); f16x2h: in addition the
# 64-channel input of the 256-channel blocks' last conv rounded to fp16 (1.54e-4 / 1.80e-4).  Both far inside the 1e-3 budget.
NET_TOL = {"fp32": 5e-6, "bf16x3": TIGHT, "f16x2": 2e-4, "f16x2h": 2.5e-4}
# inverse(forward(x)): the same deterministic F in both directions; f16x2h's fp16 intermediates / conv inputs make F a step
# function of its input, so the fp32 rounding of the recovered states is amplified (measured up to 4.2e-5 max-rel, i.e. 1 % of an
# 8-bit level)
RT_TOL = {"fp32": 5e-6, "bf16x3": 5e-6, "f16x2": 5e-6, "f16x2h": 1e-4}


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16x2", "f16x2h"])
@pytest.mark.parametrize("mode", ["photo", "art"])
def test_network_golden(golden, mode, precision):
    g = golden(f"net_{mode}")
    net, sd, sp = make_net(mode, precision)
    tol = NET_TOL[precision]
    for tag in ("16", "24x40", "32b2"):
        x = T(g[f"x_{tag}"]).cuda()
        z = net(x, forward=True)
        assert z.shape == T(g[f"z_{tag}"]).shape
        assert_close(z, T(g[f"z_{tag}"]), tol, f"{mode}.{tag} forward")
        y = net(T(g[f"zp_{tag}"]).cuda(), forward=False)
        assert_close(y, T(g[f"y_{tag}"]), tol, f"{mode}.{tag} inverse")
        rec = net(z, forward=False)
        assert_close(rec, x, RT_TOL[precision], f"{mode}.{tag} inverse(forward(x))")


@pytest.mark.parametrize("precision", ["bf16x3", "f16x2", "f16x2h"])
@pytest.mark.parametrize("mode,shape", [("photo", (1, 72, 104)), ("photo", (2, 64, 64)), ("art", (1, 40, 136)),
                                        ("photo", (1, 8, 8)), ("art", (3, 8, 12))])
def test_network_vs_oracle_ragged(mode, shape, precision):
    """tile-edge / tiny / batched shapes against the oracle on the same seeded inputs"""
    net, sd, sp = make_net(mode, precision)
    TIGHT = NET_TOL[precision]
    B, H, W = shape
    x = synthetic_frames(B, H, W, seed=11)
    with torch.no_grad():
        zr = cpu_ref.revnet_forward(x, sd, sp)
    z = net(x.cuda())
    assert_close(z, zr, TIGHT, f"{mode} {shape} forward")
    zp = zr + 0.03 * torch.randn_like(zr)
    with torch.no_grad():
        yr = cpu_ref.revnet_inverse(zp, sd, sp)
    assert_close(net(zp.cuda(), forward=False), yr, TIGHT, f"{mode} {shape} inverse")


def test_general_architectures_golden(golden):
    """RevResNet with the reference's OTHER constructor arguments (models/RevResNet.py:166-201) runs on the generic HIP ops
    (csrc/generic.hip: exact fp32): two nets against goldens minted from the reference — A: three short stages, mult 2; B: two
    stages, 1-channel input, kernel 5, a channel_reduction that pads 16 -> 64 channels per half; C: mult = 3 on 8 / 32 channels
    (2 and 10 intermediate channels: the reference floors) — plus an architecture one step
    off the published one against the oracle."""
    import ast
    from models.RevResNet import RevResNet
    g = golden("net_general")
    for tag in ("A", "B", "C"):       # (C: mult = 3 does not divide 8 or 32 channels; the reference floors, models/RevResNet.py:81)
        arch = ast.literal_eval(str(g[f"{tag}_arch"]))
        net = RevResNet(**arch)
        net.load_state_dict({k[len(tag) + 3:]: T(g[k]) for k in g.files if k.startswith(f"{tag}_w_")})
        net = net.to("cuda").eval()
        x = T(g[f"{tag}_x"]).cuda()
        z = net(x)
        assert tuple(z.shape) == tuple(g[f"{tag}_z"].shape)
        assert_close(z, T(g[f"{tag}_z"]), 2e-6, f"arch {tag} forward")
        assert_close(net(T(g[f"{tag}_zp"]).cuda(), forward=False), T(g[f"{tag}_y"]), 2e-6, f"arch {tag} inverse")
        assert_close(net(z, forward=False), x, 2e-6, f"arch {tag} inverse(forward(x))")
        with pytest.raises(RuntimeError):
            net(x[:, :, :-1])                                        # not a multiple of down_scale
        with pytest.raises(NotImplementedError):
            net.forward_u8(torch.zeros(1, 8, 8, 3, dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError):
        RevResNet(nBlocks=[1], nStrides=[1], nChannels=[4], mult=8, hidden_dim=4)
    # nBlocks = [2, 2, 2] of the published channel plan, with the cWCT in between, against the oracle
    from models.cWCT import cWCT
    arch = dict(nBlocks=[2, 2, 2], nStrides=[1, 2, 2], nChannels=[16, 64, 256], hidden_dim=16, sp_steps=2)
    net = RevResNet(**arch).to("cuda").eval()
    sd = {}
    for idx, (k, v) in enumerate(net.state_dict().items()):
        rng = np.random.Generator(np.random.PCG64([77, idx]))
        bound = 1.0 / np.sqrt(v[0].numel()) if v.dim() == 4 else 0.05
        sd[k] = T(rng.uniform(-bound, bound, size=tuple(v.shape)).astype(np.float32))
    net.load_state_dict(sd)
    xc, xs = synthetic_frames(1, 32, 48, seed=1), synthetic_frames(1, 24, 40, seed=2)
    with torch.no_grad():
        zc, zs = cpu_ref.revnet_forward(xc, sd, 2, arch), cpu_ref.revnet_forward(xs, sd, 2, arch)
        zcs = cpu_ref.transfer(zc, zs)
        ref = cpu_ref.revnet_inverse(zcs, sd, 2, 3, arch)
        got = net(cWCT().transfer(net(xc.cuda()), net(xs.cuda())), forward=False)
    assert_close(got, ref, 5e-5, "6-block net: stylised frame vs oracle")


def test_bad_shapes_raise():
    net, _, _ = make_net("photo")
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 30, 30, device="cuda"))
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 4, 4, device="cuda"))
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 4, 16, 16, device="cuda"))


def test_batch_independence():
    net, _, _ = make_net("art")
    x = synthetic_frames(3, 32, 48, seed=3).cuda()
    zb = net(x)
    for b in range(3):
        assert torch.equal(zb[b:b + 1], net(x[b:b + 1]))


# ------------------------------------------------------------------------------------------- cWCT
def test_cwct_2d_golden(golden):
    from models.cWCT import cWCT
    g = golden("cwct_2d")
    cw = cWCT()
    for N, Lp, tol in ((32, 4096, 1e-4), (128, 1024, 1e-4), (32, 50, 5e-4)):   # cond 35 / 45 / 278
        c, s = T(g[f"c_{N}_{Lp}"]).cuda(), T(g[f"s_{N}_{Lp}"]).cuda()
        assert_close(cw.whitening(c), T(g[f"whiten_{N}_{Lp}"]), tol, f"whitening {N}x{Lp}")
        assert_close(cw.coloring(T(g[f"whiten_{N}_{Lp}"]).cuda(), s), T(g[f"color_{N}_{Lp}"]), 2e-5, f"coloring {N}x{Lp}")


def test_cwct_stats_fp64(L):
    """mean/cov against an fp64 two-pass computation, large offset to stress cancellation"""
    from models.cWCT import cWCT
    cw = cWCT()
    for N, Lp in ((32, 100003), (128, 5000), (16, 777), (64, 64)):
        x = (torch.randn(N, Lp, dtype=torch.float64) * torch.linspace(0.01, 3, N, dtype=torch.float64)[:, None] + 50.0)
        st = cw.stats(x.float().cuda()).cpu()
        xd = x.float().double()
        mu = xd.mean(-1)
        cov = (xd - mu[:, None]) @ (xd - mu[:, None]).t() / (Lp - 1)
        assert st[0] == Lp
        assert float((st[1:1 + N] - mu).abs().max()) < 1e-5
        err = float((st[1 + N:].reshape(N, N) - cov).abs().max() / cov.abs().max())
        assert err < 1e-5, (N, Lp, err)


def test_cwct_transfer_golden(golden):
    from models.cWCT import cWCT
    g = golden("cwct_transfer")
    cw = cWCT()
    c, s, sb = T(g["c"]).cuda(), T(g["s"]).cuda(), T(g["s_b"]).cuda()
    assert_close(cw.transfer(c, s), T(g["transfer"]), 2e-5, "transfer")
    assert_close(cw.interpolation(c, [s], [1.0], 0.3), T(g["interp1_ac03"]), 2e-5, "interp ac=.3")
    assert_close(cw.interpolation(c, [s, sb], [0.6, 0.4], 0.0), T(g["interp2_ac0.0"]), 2e-5, "interp2 ac=0")
    assert_close(cw.interpolation(c, [s, sb], [0.6, 0.4], 0.3), T(g["interp2_ac0.3"]), 2e-5, "interp2 ac=.3")


def test_cwct_masked_golden(golden):
    from models.cWCT import cWCT
    g = golden("cwct_masked")
    cw = cWCT()
    c = T(g["c"]).cuda()
    out = cw.transfer(c, T(g["s"]).cuda(), g["cmask"], g["smask"])
    # a label may cover as few as ~50 pixels (32 channels): the per-label factor is ill-conditioned, so the
    # pointwise budget is the north_star 1e-3 while the L2 error stays two orders below
    assert_close(out, T(g["out"]), 1e-4, "transfer_seg", tol_max=TOL)
    keep = T(np.isin(g["cmask"][0], [4, 9]))
    assert torch.equal(out[0][:, keep].cpu(), T(g["c"])[0][:, keep])
    assert torch.equal(c.cpu(), T(g["c"]))          # input not mutated


def test_cwct_jitter_golden(golden):
    from models.cWCT import cWCT
    g = golden("cwct_jitter")
    cw = cWCT()
    # deterministic failures: ones(4,4) is exactly singular; diag(1,-3e-5) needs two retries.  Embed them
    # in 16x16 identity blocks (the HIP path supports N in {16,32,64,128}).
    def embed(m):
        e = torch.eye(16)
        e[:m.shape[0], :m.shape[1]] = m
        return e
    L1 = cw.cholesky_dec(embed(T(g["neg_in"])).cuda())
    assert int(cw.last_info[2]) == int(g["neg_tries"]) == 2
    assert_close(L1[:2, :2], T(g["neg_L"]), 1e-5, "chol diag(1,-3e-5)")
    L2 = cw.cholesky_dec(embed(torch.ones(4, 4)).cuda())
    assert int(cw.last_info[2]) == int(g["ones4_tries"]) == 1
    assert_close(L2[:4, :4], T(g["ones4_L"]), 1e-3, "chol ones(4,4)")
    # rank-deficient covariance (12 samples, 32 channels): same number of retries as the reference,
    # factor within the conditioning of the jittered matrix
    L3 = cw.cholesky_dec(T(g["conv"]).cuda())
    assert int(cw.last_info[2]) == int(g["tries"])
    assert_close(L3 @ L3.t(), T(g["L"]) @ T(g["L"]).t(), 1e-4, "jittered L L^T")


def test_cwct_mask_errors():
    from models.cWCT import cWCT
    cw = cWCT()
    c = torch.rand(1, 32, 8, 8, device="cuda")
    with pytest.raises(ValueError):
        cw.transfer(c, c, np.zeros((1, 4, 4), np.uint8), np.zeros((1, 8, 8), np.uint8))


# ------------------------------------------------------------------------------------------- whole stylisation
# The BASELINE-config tests run in the product default AND in the fastest opt-in mode, each with its own stated bounds:
# CFG_MODES = (precision, bound on the codes, bound on the stylised frame, bound on max|inverse(forward(x)) - x|)
CFG_MODES = [pytest.param("bf16x3", TIGHT, 2e-4, 5e-6, id="bf16x3"), pytest.param("f16x2h", 2.5e-4, 2.5e-4, 1e-4, id="f16x2h")]


@pytest.mark.parametrize("precision,tol_z,tol_y,tol_rt", CFG_MODES)
def test_config1_golden(golden, precision, tol_z, tol_y, tol_rt):
    """BASELINE config 1: photorealistic 256x256, image_transfer.py call sequence."""
    from models.cWCT import cWCT
    g = golden("config1_photo256")
    net, sd, sp = make_net("photo", precision)
    cw = cWCT(precision=precision)
    xc = synthetic_frames(1, 256, 256, seed=int(g["content_seed"])).cuda()
    xs = synthetic_frames(1, 256, 256, seed=int(g["style_seed"])).cuda()
    zc, zs = net(xc, forward=True), net(xs, forward=True)
    zcs = cw.transfer(zc, zs)
    sty = net(zcs, forward=False)
    assert_close(zc[:, :, ::8, ::8], T(g["zc_sub"]), tol_z, "z_c")
    assert_close(zcs[:, :, ::8, ::8], T(g["zcs_sub"]), max(tol_z, 2e-4), "z_cs")
    l2, mx = assert_close(sty, T(g["stylized"]), tol_y, "stylized")
    u8 = cpu_ref.to_uint8(sty.cpu())
    d = (u8.int() - T(g["stylized_u8"]).int()).abs()
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < (5e-3 if precision == "bf16x3" else 3e-2)
    stats = lambda t: np.array([float(t.min()), float(t.max()), float(t.double().mean()), float(t.double().std())])
    assert np.allclose(stats(zc), g["zc_stats"], rtol=1e-3 if precision != "bf16x3" else 1e-4, atol=1e-4 if precision != "bf16x3" else 1e-5)
    assert np.allclose(stats(zcs), g["zcs_stats"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("precision", ["bf16x3", "f16x2", "f16x2h"])
def test_masked_stylisation_vs_oracle(precision):
    """config-5 style call (per-region cWCT) at a size the oracle finishes in seconds"""
    from models.cWCT import cWCT
    net, sd, sp = make_net("photo", precision)
    cw = cWCT(precision=precision)
    H, W = 72, 128
    xc, xs = synthetic_frames(1, H, W, seed=21), synthetic_frames(1, 64, 96, seed=22)
    cm, sm = synthetic_mask(H, W, 5, seed=3)[None], synthetic_mask(64, 96, 5, seed=4, speck=False)[None]
    with torch.no_grad():
        zc, zs, zcs, sty = cpu_ref.stylize(xc, xs, sd, sp, cm, sm)
    g_zc, g_zs = net(xc.cuda()), net(xs.cuda())
    g_zcs = cw.transfer(g_zc, g_zs, cm, sm)
    tol = 2e-4 if precision == "bf16x3" else 4e-4
    assert_close(g_zcs, zcs, tol, "masked z_cs", tol_max=TOL)
    assert_close(net(g_zcs, forward=False), sty, tol, "masked stylized", tol_max=TOL)


# ------------------------------------------------------------------------------------------- full size properties
def test_full_size_properties_1024():
    """BASELINE config 2 size (photo 1024x1024): size-independent properties instead of the oracle."""
    from models.cWCT import cWCT
    net, sd, sp = make_net("photo")
    cw = cWCT()
    xc = synthetic_frames(1, 1024, 1024, seed=0).cuda()
    xs = synthetic_frames(1, 1024, 1024, seed=1).cuda()
    zc, zs = net(xc), net(xs)
    assert torch.isfinite(zc).all()
    # 1. invertibility: inverse(forward(x)) == x  (the reference holds ~3e-7)
    rec = net(zc, forward=False)
    assert float((rec - xc).abs().max()) < 5e-6
    # 2. cWCT: the transferred code has the style's mean and covariance
    zcs = cw.transfer(zc, zs)
    a = zcs[0].reshape(32, -1).double()
    b = zs[0].reshape(32, -1).double()
    cov = lambda m: (m - m.mean(-1, keepdim=True)) @ (m - m.mean(-1, keepdim=True)).t() / (m.shape[1] - 1)
    assert float((a.mean(-1) - b.mean(-1)).abs().max()) < 1e-5
    assert float((cov(a) - cov(b)).abs().max() / cov(b).abs().max()) < 1e-4
    # 3. agreement with the exact-fp32 diagnostic path on a crop-independent statistic
    net32, _, _ = make_net("photo", "fp32")
    z32 = net32(xc[:, :, :128, :128].contiguous())
    zc_crop = net(xc[:, :, :128, :128].contiguous())
    assert_close(zc_crop, z32, TIGHT, "bf16x3 vs fp32 path")
    # 4. stylised frame is finite and decodes
    sty = net(zcs, forward=False)
    assert torch.isfinite(sty).all() and sty.shape == xc.shape


# ------------------------------------------------------------------------------------------- uint8 frame edge (8(f) rank 1)
def test_uint8_frame_edge():
    """forward_u8 == forward(ToTensor(frame)); inverse_u8 == mul(255).clamp(0,255).byte() of inverse, bit for bit."""
    net, sd, sp = make_net("photo")
    g = torch.Generator().manual_seed(5)
    frames = torch.randint(0, 256, (2, 40, 72, 3), dtype=torch.uint8, generator=g)
    x = frames.permute(0, 3, 1, 2).float().div(255)               # transforms.ToTensor
    z8 = net.forward_u8(frames.cuda())
    assert torch.equal(z8, net(x.cuda()))
    with torch.no_grad():
        assert_close(z8, cpu_ref.revnet_forward(x, sd, sp), TIGHT, "forward_u8 vs oracle")
    zp = z8 * 1.7 - 0.2                                            # push some outputs outside [0,1] to hit the clamp
    y = net(zp, forward=False)
    u8 = net.inverse_u8(zp)
    assert u8.dtype == torch.uint8 and tuple(u8.shape) == (2, 40, 72, 3)
    assert torch.equal(u8.cpu(), cpu_ref.to_uint8(y.cpu()))
    assert int((u8 == 0).sum()) > 0 and int((u8 == 255).sum()) > 0
    with pytest.raises(RuntimeError):
        net.forward_u8(frames.cuda().float())


# ------------------------------------------------------------------------------------------- drop-in scripts (8(f) rank 2)
def _png(path, h, w, seed):
    from PIL import Image
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(yy * 3 + seed * 40) % 256, (xx * 2 + seed * 90) % 256, (yy + xx) % 256], -1).astype(np.uint8)
    img = (img.astype(np.int32) + rng.integers(-20, 20, img.shape)).clip(0, 255).astype(np.uint8)
    Image.fromarray(img).save(path)
    return img


def test_image_transfer_script(tmp_path):
    """image_transfer.py call sequence end to end (files in, PNG out) against the oracle on the same pixels"""
    from PIL import Image
    import image_transfer
    c = _png(tmp_path / "c.png", 50, 70, 1)                 # img_resize floors to 48 x 68
    s = _png(tmp_path / "s.png", 40, 40, 2)
    out = image_transfer.main(["--content", str(tmp_path / "c.png"), "--style", str(tmp_path / "s.png"),
                               "--out_dir", str(tmp_path / "o"), "--synthetic_weights"])
    got = np.asarray(Image.open(out))
    from utils.utils import img_resize
    ci = np.asarray(img_resize(Image.fromarray(c), 1280, 4)); si = np.asarray(img_resize(Image.fromarray(s), 1280, 4))
    assert got.shape == ci.shape == (48, 68, 3)
    sd = synthetic_state_dict(1234)
    tt = lambda a: T(np.ascontiguousarray(a)).permute(2, 0, 1)[None].float().div(255)
    with torch.no_grad():
        ref = cpu_ref.to_uint8(cpu_ref.stylize(tt(ci), tt(si), sd, 2)[3])[0].numpy()
    d = np.abs(got.astype(int) - ref.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-2


def test_video_transfer_script_sharded(tmp_path):
    """video_transfer.py on a directory of frames, two shards: every frame written once, equal to single-frame runs"""
    from PIL import Image
    import video_transfer
    fd = tmp_path / "clip"
    fd.mkdir()
    for i in range(3):
        _png(fd / f"{i:03d}.png", 32, 48, 10 + i)
    _png(tmp_path / "s.png", 36, 36, 3)
    outs = [video_transfer.main(["--video", str(fd), "--style", str(tmp_path / "s.png"), "--out_dir", str(tmp_path / "o"),
                                 "--synthetic_weights", "--shard", f"{r}/2"]) for r in range(2)]
    if os.path.isdir(outs[0]) and not any(f.endswith(".mp4") for f in os.listdir(tmp_path / "o")):
        names = sorted(os.listdir(outs[0]))
        assert names == ["00000.png", "00001.png", "00002.png"]
        import image_transfer
        single = image_transfer.main(["--content", str(fd / "001.png"), "--style", str(tmp_path / "s.png"),
                                      "--out_dir", str(tmp_path / "o1"), "--synthetic_weights"])
        assert np.array_equal(np.asarray(Image.open(os.path.join(outs[0], "00001.png"))), np.asarray(Image.open(single)))


def test_video_transfer_gpus2_end_to_end(tmp_path):
    """`video_transfer.py --gpus 2` for real (BASELINE config 5's launcher on hardware): two child processes — on a one-GPU box
    they share the GPU, on a node each gets its own through HIP_VISIBLE_DEVICES — stylise contiguous shards of a masked clip,
    the parent (which never touches the GPU) merges; the merged frames equal a single-process run bit for bit."""
    import subprocess
    import sys
    from PIL import Image
    from utils.utils import SEG_COLORS
    fd = tmp_path / "clip"
    fd.mkdir()
    for i in range(5):
        _png(fd / f"{i:03d}.png", 48, 64, 40 + i)
    _png(tmp_path / "s.png", 40, 56, 6)
    colours = np.array([c for c, _ in SEG_COLORS[:3]], dtype=np.uint8)
    Image.fromarray(colours[synthetic_mask(48, 64, 3, seed=1, speck=False)]).save(tmp_path / "cseg.png")
    Image.fromarray(colours[synthetic_mask(40, 56, 3, seed=2, speck=False)]).save(tmp_path / "sseg.png")
    base = ["--video", str(fd), "--style", str(tmp_path / "s.png"), "--synthetic_weights", "--content_seg",
            str(tmp_path / "cseg.png"), "--style_seg", str(tmp_path / "sseg.png"), "--frames_only"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video_transfer.py")]
                       + base + ["--out_dir", str(tmp_path / "o2"), "--gpus", "2"], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    import video_transfer
    single = video_transfer.main(base + ["--out_dir", str(tmp_path / "o1")])
    multi = os.path.join(str(tmp_path / "o2"), os.path.basename(single))
    names = sorted(os.listdir(single))
    assert names == sorted(os.listdir(multi)) == ["%05d.png" % i for i in range(5)]
    for n in names:
        assert np.array_equal(np.asarray(Image.open(os.path.join(single, n))), np.asarray(Image.open(os.path.join(multi, n)))), n


@pytest.mark.parametrize("precision", [None, "f16x2h"])
def test_video_transfer_script_masked(tmp_path, precision):
    """video_transfer.py with --content_seg / --style_seg (config 5's call) at the product default precision and at f16x2h: the
    branch the script takes — learnt slot count, per-label maps on the packed rows, applied inside the uint8 decode (under the
    fp16 modes: half 0 written straight into split planes) — against the oracle, frame by frame; one frame has another size."""
    from PIL import Image
    import video_transfer
    from utils.utils import SEG_COLORS, img_resize, load_segment
    fd = tmp_path / "clip"
    fd.mkdir()
    frames = [_png(fd / f"{i:03d}.png", 64, 96, 20 + i) for i in range(2)] + [_png(fd / "002.png", 48, 80, 29)]
    style = _png(tmp_path / "s.png", 56, 72, 5)
    colours = np.array([c for c, _ in SEG_COLORS[:3]], dtype=np.uint8)
    cseg = colours[synthetic_mask(64, 96, 3, seed=1, speck=False)]
    sseg = colours[synthetic_mask(56, 72, 3, seed=2, speck=False)]
    Image.fromarray(cseg).save(tmp_path / "cseg.png")
    Image.fromarray(sseg).save(tmp_path / "sseg.png")
    args = ["--video", str(fd), "--style", str(tmp_path / "s.png"), "--out_dir", str(tmp_path / "o"), "--synthetic_weights",
            "--content_seg", str(tmp_path / "cseg.png"), "--style_seg", str(tmp_path / "sseg.png"), "--frames_only"]
    out = video_transfer.main(args + (["--precision", precision] if precision else []))
    names = sorted(os.listdir(out))
    assert names == ["00000.png", "00001.png", "00002.png"]
    sd = synthetic_state_dict(1234)
    tt = lambda a: T(np.ascontiguousarray(a)).permute(2, 0, 1)[None].float().div(255)
    smask = load_segment(str(tmp_path / "sseg.png"), (72, 56))[None]
    for i, nme in enumerate(names):
        got = np.asarray(Image.open(os.path.join(out, nme)))
        assert got.shape == (64, 96, 3)                                   # the writer size comes from the first frame
        h, w = frames[i].shape[:2]
        cmask = load_segment(str(tmp_path / "cseg.png"), (w, h))[None]
        with torch.no_grad():
            sty = cpu_ref.stylize(tt(frames[i]), tt(style), sd, 2, cmask, smask)[3]
        if (h, w) != (64, 96):                                            # resized to the writer size like the script does
            sty = torch.nn.functional.interpolate(sty, size=(64, 96), mode="bicubic", align_corners=False, antialias=True)
        ref = cpu_ref.to_uint8(sty)[0].numpy()
        d = np.abs(got.astype(int) - ref.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < (1e-2 if precision is None else 5e-2), (i, d.max(), (d > 0).mean())


def test_packed_code_inplace_edits_are_not_lost():
    """a PackedCode 'is a [B,32,H,W] tensor to every caller' — also to one that writes to it.  In-place operations
    and writes through views land in the materialised tensor; the decode and the cWCT routes must then use those values, not
    the untouched packed rows."""
    from models.cWCT import cWCT
    from vstnet_amd.code import PackedCode
    net, sd, sp = make_net("photo")
    dense, _, _ = make_net("photo")
    dense.packed_code = False
    cw = cWCT()
    x, xs = synthetic_frames(1, 32, 48, seed=3).cuda(), synthetic_frames(1, 32, 48, seed=4).cuda()
    with torch.no_grad():
        zd, zs = dense(x), dense(xs)
        z = net(x)
        assert isinstance(z, PackedCode) and not z.stale
        z.mul_(1.5)
        assert z.stale and torch.equal(z.materialize(), zd * 1.5)
        assert torch.equal(net(z, forward=False), dense(zd * 1.5, forward=False))
        assert torch.equal(net.inverse_u8(z), dense.inverse_u8(zd * 1.5))
        z2 = net(x)
        z2[:, :, 4:9, 7:20] = 0.25                      # a write through a view
        ed = zd.clone()
        ed[:, :, 4:9, 7:20] = 0.25
        assert z2.stale and torch.equal(net(z2, forward=False), dense(ed, forward=False))
        assert torch.equal(cw.transfer(z2, zs), cw.transfer(ed, zs))      # statistics of the EDITED code
        z3 = net(x)
        _ = z3 + 1.0                                    # reading does not invalidate the rows
        assert not z3.stale and isinstance(cw.transfer(z3, zs), PackedCode)


def test_masked_art_mode_with_resized_masks():
    """8(f) rank 3: label maps at image resolution, artistic codes at half resolution -> NEAREST resize (upstream rule)"""
    from models.cWCT import cWCT
    net, sd, sp = make_net("art")
    cw = cWCT(resize_masks=True)
    H, W = 96, 160
    xc, xs = synthetic_frames(1, H, W, seed=31), synthetic_frames(1, 80, 112, seed=32)
    cm, sm = synthetic_mask(H, W, 3, seed=5)[None], synthetic_mask(80, 112, 3, seed=6, speck=False)[None]
    with torch.no_grad():
        zc, zs = cpu_ref.revnet_forward(xc, sd, sp), cpu_ref.revnet_forward(xs, sd, sp)
        cmr = cw.resize(cm[0], H // 2, W // 2)[None]
        smr = cw.resize(sm[0], 40, 56)[None]
        ref = cpu_ref.transfer_seg(zc, zs, cmr, smr)
    got = cw.transfer(net(xc.cuda()), net(xs.cuda()), cm, sm)
    assert_close(got, ref, 2e-3, "masked art z_cs", tol_max=2e-2)       # 128 channels, ~200-pixel regions: ill-conditioned
    with pytest.raises(ValueError):
        cWCT().transfer(net(xc.cuda()), net(xs.cuda()), cm, sm)


# ------------------------------------------------------------------------------------------- more surface coverage
def test_cached_style_equals_transfer_and_batch2_masked():
    from models.cWCT import cWCT
    net, sd, sp = make_net("photo")
    cw = cWCT()
    xc = synthetic_frames(2, 40, 56, seed=41).cuda()
    xs = synthetic_frames(2, 32, 48, seed=42).cuda()
    zc, zs = net(xc), net(xs)
    ref = cw.transfer(zc, zs)
    got = cw.transfer_with_stats(zc, cw.style_stats(zs))              # prefactored style (Cholesky cached)
    assert_close(got, ref, 1e-6, "transfer_with_stats vs transfer")
    one = cw.transfer_with_stats(zc, cw.style_stats(zs[:1]))          # one style for every frame of the batch
    # (zs[:1] is a dense tensor: its statistics come from the NCHW kernel, ref's from the packed-row kernel - two summation orders)
    assert_close(one[0], ref[0], 5e-6, "single cached style, sample 0")
    # masked, batch of 2 with different masks per sample, against the oracle
    cm = np.stack([synthetic_mask(40, 56, 3, seed=7), synthetic_mask(40, 56, 4, seed=8)])
    sm = np.stack([synthetic_mask(32, 48, 3, seed=9, speck=False), synthetic_mask(32, 48, 4, seed=10, speck=False)])
    with torch.no_grad():
        r = cpu_ref.transfer_seg(zc.cpu(), zs.cpu(), cm, sm)
    assert_close(cw.transfer(zc, zs, cm, sm), r, 2e-4, "batch-2 masked transfer", tol_max=TOL)


def test_input_forms_and_sample():
    """non-contiguous / fp64 inputs are accepted like the reference's modules accept them; sample() runs"""
    from models.cWCT import cWCT
    net, sd, sp = make_net("photo")
    x = synthetic_frames(1, 32, 32, seed=51)
    base = net(x.cuda())
    xt = x.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)      # same values, non-contiguous strides
    assert not xt.is_contiguous()
    assert torch.equal(net(xt.cuda()), base)
    assert_close(net(x.double().cuda()), base, 1e-7, "fp64 input")
    assert cWCT(use_double=True).use_double is True  # implemented: test_cwct_use_double_golden
    out = cWCT().transfer(base.double(), base.double().flip(-1))
    assert out.dtype == torch.float64 and out.shape == base.shape
    xc, xs, xcs, cyc = net.sample(cWCT(), x, synthetic_frames(1, 32, 32, seed=52), "cuda")
    assert xcs.shape == x.shape and cyc.shape == x.shape and torch.isfinite(xcs).all()


def test_cwct_use_double_golden(golden):
    """cWCT(use_double=True) (models/cWCT.py:13-16,35-47,66,106,220,238,259) against goldens minted from the reference with the
    flag set, on a content code with cond(cov) ~ 1e6: the fp64 path must land on the reference's fp64 result to fp32 output
    rounding, where fp32 arithmetic (the reference's own fp32 path included: 7e-3 .. 8e-3, recorded in the fixture) cannot."""
    from models.cWCT import cWCT
    g = golden("cwct_double")
    c, s1, s2 = T(g["c"]).cuda(), T(g["s1"]).cuda(), T(g["s2"]).cuda()
    cwd, cw32 = cWCT(use_double=True), cWCT(precision="fp32")
    for ac in (0.0, 0.3):
        ref = T(g[f"interp_ac{ac}"])
        out = cwd.interpolation(c, [s1, s2], [0.7, 0.3], ac)
        assert out.dtype == torch.float32 and out.shape == c.shape
        assert_close(out, ref, 2e-6, f"use_double interpolation alpha_c={ac}")
        e32 = rel_err(cw32.interpolation(c, [s1, s2], [0.7, 0.3], ac), ref)[1]
        assert e32 > 1e-4 and float(g[f"fp32_path_max_rel_ac{ac}"]) > 1e-3      # the case separates fp32 from fp64 arithmetic
    out = cwd.transfer(c.clone(), s1, g["cmask"], g["smask"])
    assert_close(out, T(g["masked"]), 2e-6, "use_double transfer_seg")
    assert_close(cwd.transfer(c, s1), cpu_ref.transfer(T(g["c"]), T(g["s1"]), use_double=True), 2e-6, "use_double transfer")
    # the fp64 jitter branch (csrc/cwct64.hip: same failure rule, eps rounded to float32 like the reference's float32 identity,
    # models/cWCT.py:120-124) on a code with a constant channel: one retry, golden minted from the reference
    cj, sj = T(g["c_jit"]).cuda(), T(g["s_jit"]).cuda()
    assert_close(cwd.interpolation(cj, [sj], [1.0], 0.0), T(g["interp_jit"]), 2e-6, "use_double jitter branch")
    # packed codes are materialised, the 2-D helpers run in fp64 too
    net, sd, sp = make_net("photo")
    z, zs = net(synthetic_frames(1, 32, 48, seed=1).cuda()), net(synthetic_frames(1, 32, 48, seed=2).cuda())
    from vstnet_amd.code import PackedCode
    got = cwd.transfer(z, zs)
    assert isinstance(z, PackedCode) and not isinstance(got, PackedCode)
    assert_close(got, cpu_ref.transfer(z.materialize().cpu(), zs.materialize().cpu(), use_double=True), 2e-6, "use_double on a packed code")
    x2 = T(g["c"])[0].reshape(32, -1)
    assert_close(cwd.whitening(x2.cuda()), cpu_ref.whitening(x2.double()).float(), 5e-6, "use_double whitening (cond 1e6)")


def test_cwct_every_route_vs_oracle():
    """each entry of cWCT.ROUTES, taken on purpose, against the oracle — `last_route` says which one ran"""
    from models.cWCT import cWCT
    from vstnet_amd.code import PackedCode
    rng = np.random.default_rng(5)
    took = set()

    def codes(N, h, w, hs, ws):
        return (T(rng.standard_normal((1, N, h, w)).astype(np.float32)) * 0.7 + 0.2,
                T(rng.standard_normal((1, N, hs, ws)).astype(np.float32)) * 1.2 - 0.1)
    # dense, every N
    for N in (16, 32, 64, 128):
        c, s = codes(N, 24, 40, 20, 36)
        cw = cWCT()
        assert_close(cw.transfer(c.cuda(), s.cuda()), cpu_ref.transfer(c, s), 2e-4, f"dense N={N}", tol_max=TOL)
        assert cw.last_route == "dense"
        took.add(cw.last_route)
    # masked: single pass (N = 32, 128), per label (N = 16)
    cm, sm = synthetic_mask(24, 40, 3, seed=1)[None], synthetic_mask(20, 36, 3, seed=2, speck=False)[None]
    for N, want in ((32, "masked_single_pass"), (16, "masked_per_label")):
        c, s = codes(N, 24, 40, 20, 36)
        cw = cWCT()
        assert_close(cw.transfer(c.cuda(), s.cuda(), cm, sm), cpu_ref.transfer_seg(c, s, cm, sm), 5e-4, want, tol_max=5e-3)
        assert cw.last_route == want
        took.add(want)
    # fp64, unmasked and masked
    c, s = codes(32, 24, 40, 20, 36)
    cwd = cWCT(use_double=True)
    assert_close(cwd.transfer(c.cuda(), s.cuda()), cpu_ref.transfer(c, s, use_double=True), 2e-6, "dense_f64")
    took.add(cwd.last_route)
    assert_close(cwd.transfer(c.cuda(), s.cuda(), cm, sm), cpu_ref.transfer_seg(c, s, cm, sm, use_double=True), 2e-6, "masked f64")
    took.add(cwd.last_route)
    # packed rows, unmasked and masked (photorealistic code from the network)
    net, sd, sp = make_net("photo")
    xc, xs = synthetic_frames(1, 24, 40, seed=1), synthetic_frames(1, 24, 40, seed=2)
    cw = cWCT()
    with torch.no_grad():
        z, zs = net(xc.cuda()), net(xs.cuda())
        t = cw.transfer(z, zs)
        assert isinstance(t, PackedCode) and cw.last_route == "packed_rows"
        took.add(cw.last_route)
        assert_close(t.materialize(), cpu_ref.transfer(z.materialize().cpu(), zs.materialize().cpu()), 2e-5, "packed_rows")
        cm2, sm2 = synthetic_mask(24, 40, 3, seed=3)[None], synthetic_mask(24, 40, 3, seed=4, speck=False)[None]
        plan = cw.learn_slots(cw.plan_masks(cm2, sm2, z.shape, zs.shape, z.device))
        tm = cw.transfer_with_plan(z, zs, plan)
        assert isinstance(tm, PackedCode) and cw.last_route == "masked_packed_rows"
        took.add(cw.last_route)
        assert_close(tm.materialize(), cpu_ref.transfer_seg(z.materialize().cpu(), zs.materialize().cpu(), cm2, sm2), 2e-4,
                     "masked_packed_rows", tol_max=TOL)
    assert took == set(cWCT.ROUTES), set(cWCT.ROUTES) - took


def test_native_runner_matches_python_path(tmp_path):
    """tools/vst_run (C++ host, C ABI only — no Python / torch in the process) == the drop-in classes, bit for bit"""
    import subprocess
    from models.cWCT import cWCT
    from vstnet_amd.export import export_state_dict
    if not os.path.exists(_lib.RUNNER_BIN):
        _lib.build_runner()
    for mode, precision in (("photo", "f16x2h"), ("art", "f16x2h"), ("photo", "f16x2"), ("photo", "bf16x3")):
        net, sd, sp = make_net(mode, precision)
        hd = 16 if mode == "photo" else 64
        export_state_dict(sd, str(tmp_path / "w.bin"), hd, sp)
        g = torch.Generator().manual_seed(9)
        c = torch.randint(0, 256, (1, 40, 64, 3), dtype=torch.uint8, generator=g)
        s = torch.randint(0, 256, (1, 32, 48, 3), dtype=torch.uint8, generator=g)
        c.numpy().tofile(tmp_path / "c.rgb"); s.numpy().tofile(tmp_path / "s.rgb")
        r = subprocess.run([_lib.RUNNER_BIN, str(tmp_path / "w.bin"), str(tmp_path / "c.rgb"), "40", "64",
                            str(tmp_path / "s.rgb"), "32", "48", str(tmp_path / "o.rgb"), precision],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr + r.stdout
        got = np.fromfile(tmp_path / "o.rgb", dtype=np.uint8).reshape(40, 64, 3)
        cw = cWCT(precision=precision)
        zc, zs = net.forward_u8(c.cuda()), net.forward_u8(s.cuda())
        ref = net.inverse_u8(cw.transfer_with_stats(zc, cw.style_stats(zs)))[0].cpu().numpy()
        assert np.array_equal(got, ref), (mode, precision)


def test_lab_luminance_postprocess(L, golden):
    """vst_lab_luminance against the golden minted from the fork's color.py and the oracle
    (fp32 pointwise; pow() differs by an ulp or two between libm and the device -> 1e-5 absolute on [0,1])."""
    from vstnet_amd.color import luminance_transfer
    g = golden("lab")
    c, s = T(g["content"]), T(g["stylized"])
    got = luminance_transfer(c.cuda(), s.cuda()).cpu()
    assert float((got - T(g["out"])).abs().max()) < 1e-5
    # unclamped decoder output, odd sizes (scalar path), batch > 1, in-place over the stylised image
    for (B, H, W, seed) in ((1, 7, 13, 3), (3, 33, 50, 4), (2, 64, 64, 5)):
        c = synthetic_frames(B, 8 * ((H + 7) // 8), 8 * ((W + 7) // 8), seed=seed)[:, :, :H, :W].contiguous()
        s = (synthetic_frames(B, 8 * ((H + 7) // 8), 8 * ((W + 7) // 8), seed=seed + 50)[:, :, :H, :W] * 1.6 - 0.3).contiguous()
        ref = cpu_ref.luminance_transfer(c, s)
        sg = s.cuda()
        got = luminance_transfer(c.cuda(), sg, out=sg).cpu()
        assert got.shape == ref.shape and float((got - ref).abs().max()) < 1e-5, (B, H, W)
    # exact edge values: black, white, the sRGB knee
    c = torch.tensor([0.0, 1.0, 0.04045, 0.5]).view(1, 1, 1, 4).expand(1, 3, 1, 4).contiguous()
    s = torch.tensor([1.0, 0.0, 0.04045, 0.25]).view(1, 1, 1, 4).expand(1, 3, 1, 4).contiguous()
    assert float((luminance_transfer(c.cuda(), s.cuda()).cpu() - cpu_ref.luminance_transfer(c, s)).abs().max()) < 1e-5
    with pytest.raises(RuntimeError):
        luminance_transfer(c, s)                                        # no CPU path
    assert L.vst_lab_luminance(None, None, None, 1, 4, 4, None) == -1


def test_image_transfer_script_preserve_luminance(tmp_path):
    from PIL import Image
    import image_transfer
    c = _png(tmp_path / "c.png", 48, 68, 11)
    s = _png(tmp_path / "s.png", 40, 40, 12)
    out = image_transfer.main(["--content", str(tmp_path / "c.png"), "--style", str(tmp_path / "s.png"),
                               "--out_dir", str(tmp_path / "o"), "--synthetic_weights", "--preserve_luminance"])
    got = np.asarray(Image.open(out))
    sd = synthetic_state_dict(1234)
    tt = lambda a: T(np.ascontiguousarray(a)).permute(2, 0, 1)[None].float().div(255)
    with torch.no_grad():
        sty = cpu_ref.stylize(tt(c), tt(s), sd, 2)[3]
        ref = cpu_ref.to_uint8(cpu_ref.luminance_transfer(tt(c), sty))[0].numpy()
    d = np.abs(got.astype(int) - ref.astype(int))
    assert got.shape == ref.shape and d.max() <= 1 and (d > 0).mean() < 2e-2


def test_frame_pipeline_matches_sequential():
    """the overlapped pinned-buffer frame loop returns exactly what the one-frame-at-a-time
    loop returns (plain, masked, and with a decode hook that writes another size), in order."""
    from models.cWCT import cWCT
    from vstnet_amd.pipeline import FramePipeline, AsyncSink, prefetch
    net, sd, sp = make_net("photo")
    cw = cWCT()
    H, W, N = 64, 96, 9
    frames = [(synthetic_frames(1, H, W, seed=100 + i)[0].permute(1, 2, 0) * 255).byte().numpy() for i in range(N)]
    style = (synthetic_frames(1, 48, 64, seed=7)[0].permute(1, 2, 0) * 255).byte()[None].cuda()
    cmask, smask = synthetic_mask(H, W, 3, seed=1)[None], synthetic_mask(48, 64, 3, seed=2)[None]
    with torch.no_grad():
        z_s = net.forward_u8(style)
        stats = cw.style_stats(z_s)
        for masked in (False, True):
            tf = (lambda z, i: cw.transfer(z, z_s, cmask, smask)) if masked else (lambda z, i: cw.transfer_with_stats(z, stats))
            ref = [net.inverse_u8(tf(net.forward_u8(T(f)[None].cuda()), i))[0].cpu().numpy() for i, f in enumerate(frames)]
            got = {}
            sink = AsyncSink(lambda i, a: got.__setitem__(i, a))
            pipe = FramePipeline(net, tf, H, W, depth=3, compute_streams=2)
            assert pipe.run(prefetch(iter(frames), ahead=2), sink, start_index=5) == N
            sink.close()
            assert sorted(got) == list(range(5, 5 + N))
            for i in range(N):
                assert np.array_equal(got[5 + i], ref[i]), (masked, i)
        # decode hook: half-size output
        def decode(z):
            y = net(z, forward=False)[:, :, ::2, ::2]
            return y.mul(255).clamp(0, 255).byte().permute(0, 2, 3, 1).contiguous()
        tf = lambda z, i: cw.transfer_with_stats(z, stats)
        ref = [decode(tf(net.forward_u8(T(f)[None].cuda()), 0))[0].cpu().numpy() for f in frames[:4]]
        out = []
        pipe = FramePipeline(net, tf, H, W, depth=2, compute_streams=1, decode=decode, out_height=H // 2, out_width=W // 2)
        pipe.run(frames[:4], lambda i, a: out.append(a.copy()))
        assert all(np.array_equal(a, b) for a, b in zip(out, ref))
        with pytest.raises(ValueError):
            pipe.run([frames[0][:32]], lambda i, a: None)
    with pytest.raises(ValueError):
        FramePipeline(net, tf, 30, 32)



# ------------------------------------------------------------------------------------------- BASELINE configs 3-5 at full size
def _cov(m):
    d = m - m.mean(-1, keepdim=True)
    return d @ d.t() / (m.shape[1] - 1)


@pytest.mark.parametrize("precision,tol_z,tol_y,tol_rt", CFG_MODES)
def test_full_size_config3_art_batch4(precision, tol_z, tol_y, tol_rt):
    """config 3's per-GPU share: artistic mode, 4 frames of 1024x1024 in one batch.  Properties: every frame of the
    batch equals the same frame run alone (bit for bit), the pass inverts, the code takes the style's moments."""
    from models.cWCT import cWCT
    net, sd, sp = make_net("art", precision)
    cw = cWCT(precision=precision)
    x = synthetic_frames(4, 1024, 1024, seed=3).cuda()
    xs = synthetic_frames(1, 1024, 1024, seed=1).cuda()
    with torch.no_grad():
        z = net(x)
        assert z.shape == (4, 128, 512, 512)
        z1 = net(x[2:3])
        assert torch.equal(z[2:3], z1)
        assert float((net(z, forward=False) - x).abs().max()) < tol_rt
        stats = cw.style_stats(net(xs))
        zcs = cw.transfer_with_stats(z, stats)
        zs = net(xs)[0].reshape(128, -1).double()
        for b in (0, 3):
            a = zcs[b].reshape(128, -1).double()
            assert float((a.mean(-1) - zs.mean(-1)).abs().max()) < 1e-5 * max(1.0, float(zs.mean(-1).abs().max()))
            assert float((_cov(a) - _cov(zs)).abs().max() / _cov(zs).abs().max()) < 1e-4
        sty = net(zcs, forward=False)
        assert torch.isfinite(sty).all() and sty.shape == x.shape


# Large frames against the ORACLE, both ends of the image and both directions (VERDICT r3 "what's weak" 1): the oracle runs on
# a CROP x CROP crop that shares the image's real top-left (or bottom-right) border, and the CORNER x CORNER corner next to that
# border is compared.  A pass reaches at most 10*3*1 + 10*3*2 + 12*3*4 = 234 full-resolution pixels (30 + 2 blocks x 3 convs,
# one pixel each at the block's resolution), less than the CROP - CORNER = 256 pixels between the corner and the crop's
# artificial borders, so the corner sees exactly what it sees in the full frame - including the ReflectionPad at the real
# border (models/RevResNet.py:79-88), the ragged last tile row / column and the far tiles' addresses.
CROP, CORNER = 768, 512


def _corner(H, W, which):
    """(crop slices in the frame, corner slices inside the crop, corner slices in the frame)"""
    if which == "tl":
        return (slice(0, CROP), slice(0, CROP)), (slice(0, CORNER), slice(0, CORNER)), (slice(0, CORNER), slice(0, CORNER))
    return ((slice(H - CROP, H), slice(W - CROP, W)), (slice(CROP - CORNER, CROP), slice(CROP - CORNER, CROP)),
            (slice(H - CORNER, H), slice(W - CORNER, W)))


def check_corners_vs_oracle(x_cpu, z_c, z_cs, sty, sd, sp, tol_z, tol_y, what):
    """encode: the oracle's code of the crop of x vs the GPU's code; decode: the oracle's revnet_inverse (models/RevResNet.py:
    225-239) of the crop of the GPU's OWN transferred code vs the GPU's stylised frame - top-left and bottom-right."""
    H, W = x_cpu.shape[2:]
    torch.set_num_threads(16)
    with torch.no_grad():
        for which in ("tl", "br"):
            (cy, cx), (iy, ix), (fy, fx) = _corner(H, W, which)
            z_or = cpu_ref.revnet_forward(x_cpu[:, :, cy, cx].contiguous(), sd, sp)
            assert_close(z_c[:, :, fy, fx], z_or[:, :, iy, ix], tol_z, f"{what}: code, {which} corner vs oracle")
            y_or = cpu_ref.revnet_inverse(z_cs[:, :, cy, cx].float().cpu().contiguous(), sd, sp)
            assert_close(sty[:, :, fy, fx], y_or[:, :, iy, ix], tol_y, f"{what}: decoded frame, {which} corner vs oracle")


@pytest.mark.parametrize("precision,tol_z,tol_y,tol_rt", CFG_MODES)
def test_full_size_config4_4096(precision, tol_z, tol_y, tol_rt):
    """config 4: one 4096x4096 photorealistic image on one GPU (2 GiB of state per half; the pass runs in sub-batches
    sized for the Infinity Cache).  Code AND decoded frame against the ORACLE at the top-left and the bottom-right corner
    (check_corners_vs_oracle: the far tiles, byte offsets near 2^30), then properties: invertibility, crop consistency on the
    GPU, style moments."""
    from models.cWCT import cWCT
    net, sd, sp = make_net("photo", precision)
    cw = cWCT(precision=precision)
    x_cpu = synthetic_frames(1, 4096, 4096, seed=11)
    x = x_cpu.cuda()
    xs = synthetic_frames(1, 1024, 1024, seed=1).cuda()
    with torch.no_grad():
        z = net(x)
        zcs_ = cw.transfer(z, net(xs))
        sty_ = net(zcs_, forward=False)
        check_corners_vs_oracle(x_cpu, z, zcs_, sty_, sd, sp, tol_z, TIGHT if precision == "bf16x3" else tol_y,
                                f"4096x4096 ({precision})")
        del zcs_, sty_
        assert float((net(z, forward=False) - x).abs().max()) < tol_rt
        zc = net(x[:, :, :1024, :1024].contiguous())
        d = (z[:, :, :512, :512] - zc[:, :, :512, :512]).abs().max()
        assert float(d) < 1e-4 * float(zc.abs().max())           # same arithmetic, different tile grid -> fp32 noise only
        zs = net(xs)
        zcs = cw.transfer(z, zs)
        a, b = zcs[0].reshape(32, -1).double(), zs[0].reshape(32, -1).double()
        assert float((a.mean(-1) - b.mean(-1)).abs().max()) < 1e-5
        assert float((_cov(a) - _cov(b)).abs().max() / _cov(b).abs().max()) < 1e-4
        del zcs, a
        out = net.inverse_u8(cw.transfer(z, zs))
        assert out.shape == (1, 4096, 4096, 3) and out.dtype == torch.uint8


@pytest.mark.parametrize("precision", ["bf16x3", "f16x2h"])
@pytest.mark.parametrize("kind", ["bands", "noise"])
def test_full_size_config5_1080p_masked(kind, precision):
    """config 5's frame: 1920x1080, 5-label masks (+ a 6-pixel speck that must stay untouched) — vertical bands, and the
    worst case for anything per-label: an independent random label per pixel.  Property: inside every valid label the
    transferred code has the moments of the style code inside the same label; the speck keeps the content code."""
    from models.cWCT import cWCT
    net, sd, sp = make_net("photo", precision)
    cw = cWCT(precision=precision)
    H, W = 1080, 1920
    x = synthetic_frames(1, H, W, seed=21).cuda()
    xs = synthetic_frames(1, H, W, seed=22).cuda()
    cm, sm = synthetic_mask(H, W, 5, seed=3, kind=kind)[None], synthetic_mask(H, W, 5, seed=4, speck=False, kind=kind)[None]
    with torch.no_grad():
        zc, zs = net(x), net(xs)
        zc0 = zc.clone()
        zcs = cw.transfer(zc, zs, cm, sm)
    cmt, smt = T(cm[0]).cuda().reshape(-1), T(sm[0]).cuda().reshape(-1)
    a_all, b_all, c_all = zcs[0].reshape(32, -1), zs[0].reshape(32, -1), zc0[0].reshape(32, -1)
    for label in range(5):
        a, b = a_all[:, cmt == label].double(), b_all[:, smt == label].double()
        assert a.shape[1] > 10 and b.shape[1] > 10
        assert float((a.mean(-1) - b.mean(-1)).abs().max()) < 2e-5 * max(1.0, float(b.mean(-1).abs().max()))
        assert float((_cov(a) - _cov(b)).abs().max() / _cov(b).abs().max()) < 2e-4, label
    speck = cmt == 5
    assert int(speck.sum()) == 6 and torch.equal(a_all[:, speck], c_all[:, speck])
    # the encode and the decode against the ORACLE at both corners (check_corners_vs_oracle), the masked cWCT against the
    # oracle's transfer_seg fed with the same codes, and the route video_transfer.py takes
    # (learnt slot count -> per-label maps on the packed rows, applied inside the uint8 decode) against the dense route
    tol_z = TIGHT if precision == "bf16x3" else 2.5e-4
    with torch.no_grad():
        torch.set_num_threads(16)
        x_cpu = x.cpu()
        # 1080 rows = 270 quarter-resolution rows = 16 tiles + a 14-row ragged tile: the bottom-right corner covers it
        check_corners_vs_oracle(x_cpu, zc0, zcs, net(zcs, forward=False), sd, sp, tol_z, tol_z, f"1080p {kind} ({precision})")
        ref = cpu_ref.transfer_seg(zc0.cpu(), zs.cpu(), cm, sm)
        assert_close(zcs, ref, 2e-4, f"1080p masked cWCT vs oracle ({kind})", tol_max=TOL)
        plan = cw.bind_style(cw.learn_slots(cw.plan_masks(cm, sm, zc.shape, zs.shape, zc.device)), zs)
        t = cw.transfer_with_plan(net(x), None, plan)
        from vstnet_amd.code import PackedCode
        assert isinstance(t, PackedCode) and t.pending_labels is not None
        u8_packed, u8_dense = net.inverse_u8(t), net.inverse_u8(zcs)
        dd = (u8_packed.int() - u8_dense.int()).abs()
        assert int(dd.max()) <= 1 and float((dd > 0).float().mean()) < (1e-3 if precision == "bf16x3" else 2e-2)


@pytest.mark.parametrize("mode,batch,shape", [("photo", 1, (1024, 1024)), ("photo", 1, (200, 280)), ("photo", 1, (72, 40)),
                                              ("art", 3, (136, 104)), ("photo", 2, (1080, 360))])
def test_stage3_lean_option_is_bit_identical(mode, batch, shape):
    """VST_OPT_STAGE3_LEAN (vstnet.h): the 256-channel convs of bf16x3 as half-CU workgroups (4 waves, 8 x 16 tiles) give the
    same bits as the 8-wave form (same MFMA order per accumulator) - full size, ragged tiles, a frame smaller than a tile, an
    artistic batch (several images per launch), a tall batch whose quarter-resolution height (270) is not a multiple of 8."""
    from models.cWCT import cWCT
    from vstnet_amd import _lib
    net, sd, sp = make_net(mode, "bf16x3")
    cw = cWCT(precision="bf16x3")
    h, w = shape
    xc, xs = synthetic_frames(batch, h, w, seed=0).cuda(), synthetic_frames(batch, h, w, seed=1).cuda()
    res = []
    assert _lib.get_option(_lib.OPT_STAGE3_LEAN) in (0, 1)
    before = _lib.get_option(_lib.OPT_STAGE3_LEAN)
    try:
        with torch.no_grad():
            for lean in (0, 1):
                _lib.set_option(_lib.OPT_STAGE3_LEAN, lean)
                zc = net(xc)
                sty = net(cw.transfer(zc, net(xs)), forward=False)
                res.append((torch.as_tensor(zc).float().clone(), sty.clone()))
    finally:
        _lib.set_option(_lib.OPT_STAGE3_LEAN, before)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    with pytest.raises(_lib.VstError):
        _lib.set_option(99, 1)


# ------------------------------------------------------------------------------------------- full size vs the oracle itself
_FULL = {}


def _oracle_1024(mode):
    """cpu_ref.stylize once per mode at 1024x1024 (config 2 / one frame of config 3): ~6 s on the box's 16 cores."""
    if mode not in _FULL:
        hd, sp = (16, 2) if mode == "photo" else (64, 1)
        sd = synthetic_state_dict(1234, hd, sp)
        torch.set_num_threads(16)
        xc, xs = synthetic_frames(1, 1024, 1024, seed=0), synthetic_frames(1, 1024, 1024, seed=1)
        with torch.no_grad():
            _FULL[mode] = (xc, xs) + tuple(cpu_ref.stylize(xc, xs, sd, sp))
    return _FULL[mode]


@pytest.mark.parametrize("precision,tol", [("f16x2h", 2.5e-4), ("f16x2", 2e-4), ("bf16x3", TIGHT)])
@pytest.mark.parametrize("mode", ["photo", "art"])
def test_full_size_1024_vs_oracle(mode, precision, tol):
    """BASELINE config 2 (photo) and a config-3 frame (art) at the full 1024x1024 against the oracle on the same inputs:
    rel-L2 and max-rel of z_c, z_cs and the stylised frame (north_star budget 1e-3; the asserted bounds are tighter)."""
    from models.cWCT import cWCT
    xc, xs, zc, zs, zcs, sty = _oracle_1024(mode)
    net, _, _ = make_net(mode, precision)
    cw = cWCT()
    with torch.no_grad():
        g_zc, g_zs = net(xc.cuda()), net(xs.cuda())
        g_zcs = cw.transfer(g_zc, g_zs)
        g_sty = net(g_zcs, forward=False)
    assert_close(g_zc, zc, tol, f"{mode}/{precision} z_c")
    assert_close(g_zcs, zcs, tol, f"{mode}/{precision} z_cs", tol_max=max(tol, 2e-4))
    assert_close(g_sty, sty, tol, f"{mode}/{precision} stylized")
    assert tol <= TOL


def test_batch_coupled_jitter_golden(golden):
    """the reference factors the [B,N,N] stack at once: sample 0's singular covariance jitters sample 1 too (cWCT.py:122-128)"""
    from models.cWCT import cWCT
    g = golden("cwct_batch_jitter")
    cw = cWCT()
    c, s = T(g["c"]).cuda(), T(g["s"]).cuda()
    for ac in (0.0, 0.3):
        out = cw.interpolation(c, [s], [1.0], ac)
        # sample 0 has a zero-variance channel: its whitening is ill-conditioned by construction (1/sqrt(eps) gain)
        assert_close(out[1], T(g[f"out_ac{ac}"])[1], 2e-5, f"coupled sample, alpha_c={ac}", tol_max=1e-4)
        assert_close(out[0], T(g[f"out_ac{ac}"])[0], 2e-3, f"singular sample, alpha_c={ac}", tol_max=2e-2)
    assert cw.last_info.cpu().tolist()[0] == int(g["tries"]) == 1
    # per-sample (uncoupled) factoring of sample 1 differs measurably: the coupling is what the golden pins
    alone = cw.interpolation(c[1:2], [s[1:2]], [1.0], 0.0)
    assert rel_err(alone[0], T(g["out_ac0.0"])[1])[0] > 1e-5


def test_art_mode_inplace_and_masked_128():
    """N = 128 paths with thin coverage so far: the in-place split-operand apply (y == x) and the masked apply on a
    >= 2000-pixel region, pointwise against the oracle"""
    from models.cWCT import cWCT
    net, sd, sp = make_net("art")
    net.packed_code = False                           # the dense [B,128,H/2,W/2] route (packed codes: test_packed_code_artistic)
    cw = cWCT(resize_masks=True)
    H, W = 192, 256                                   # code 96 x 128 = 12288 pixels (L % 64 == 0: split-operand kernel)
    xc, xs = synthetic_frames(1, H, W, seed=71), synthetic_frames(1, H, W, seed=72)
    with torch.no_grad():
        zc_o, zs_o = cpu_ref.revnet_forward(xc, sd, sp), cpu_ref.revnet_forward(xs, sd, sp)
        ref = cpu_ref.transfer(zc_o, zs_o)
        zc, zs = net(xc.cuda()), net(xs.cuda())
        stats = cw.style_stats(zs)
        out = cw.transfer_with_stats(zc, stats)
        assert_close(out, ref, 2e-4, "art transfer (N=128)", tol_max=TOL)
        keep = zc.clone()
        same = cw.transfer_with_stats(zc, stats, inplace=True)
        assert same.data_ptr() == zc.data_ptr() and torch.equal(same, out) and not torch.equal(zc, keep)
        # masked: two labels of ~6000 code pixels each (>> 128 channels: well conditioned), masks at image resolution
        cm = np.zeros((1, H, W), np.uint8); cm[:, :, W // 2:] = 1
        sm = np.zeros((1, H, W), np.uint8); sm[:, H // 2:, :] = 1
        cmr, smr = cw.resize(cm[0], H // 2, W // 2)[None], cw.resize(sm[0], H // 2, W // 2)[None]
        refm = cpu_ref.transfer_seg(zc_o, zs_o, cmr, smr)
        gotm = cw.transfer(keep, zs, cm, sm)
        assert_close(gotm, refm, 2e-4, "masked art z_cs, 6144-pixel regions", tol_max=TOL)


def test_cwct_apply_precision_knob_ill_conditioned():
    """N = 128, L % 64 == 0 (the split-operand bf16 apply) on a covariance of condition number ~1e4, where T = Ls Lc^-1 has
    large cancelling entries: the split path must stay as close to the fp64 oracle as the exact-fp32 path does, and
    precision='fp32' must select the exact kernels (the knob used to stop at RevResNet)."""
    from models.cWCT import cWCT
    N, Lp = 128, 64 * 64
    g = torch.Generator().manual_seed(5)
    q, _ = torch.linalg.qr(torch.randn(N, N, generator=g, dtype=torch.float64))
    sv = torch.logspace(0, -2, N, dtype=torch.float64)            # singular values 1 .. 1e-2 -> cond(cov) = 1e4
    c = (q @ (sv[:, None] * torch.randn(N, Lp, generator=g, dtype=torch.float64)) + 0.3).float().reshape(1, N, 64, 64)
    s = (torch.randn(N, N, generator=g, dtype=torch.float64) @ torch.randn(N, Lp, generator=g, dtype=torch.float64) * 0.1).float()
    s = s.reshape(1, N, 64, 64)
    cc = c.reshape(N, -1).double()
    cc = cc - cc.mean(-1, keepdim=True)
    assert 5e3 < float(torch.linalg.cond(cc @ cc.t() / (Lp - 1))) < 5e4
    ref = cpu_ref.transfer(c.double(), s.double())                # fp64 oracle
    out_split = cWCT(precision="bf16x3").transfer(c.cuda(), s.cuda())
    out_exact = cWCT(precision="fp32").transfer(c.cuda(), s.cuda())
    e_split, e_exact = rel_err(out_split, ref), rel_err(out_exact, ref)
    assert not torch.equal(out_split, out_exact)                  # the knob really switches kernels
    assert e_exact[0] < 1e-3 and e_exact[1] < 1e-3, e_exact      # fp32 statistics + Cholesky at cond 1e4
    assert e_split[0] <= max(2 * e_exact[0], 2e-4) and e_split[1] <= max(2 * e_exact[1], 5e-4), (e_split, e_exact)
    assert_close(out_split, out_exact, 2e-4, "split vs exact apply", tol_max=5e-4)


def test_network_vs_oracle_seeded_shapes():
    """a seeded sweep of odd frame shapes (tile edges in every position, batches, both modes) against the oracle"""
    rng = np.random.default_rng(2024)
    nets = {m: make_net(m) for m in ("photo", "art")}
    for case in range(10):
        mode = "photo" if case % 3 else "art"
        net, sd, sp = nets[mode]
        B = int(rng.integers(1, 4))
        H, W = (int(4 * rng.integers(2, 41)), int(4 * rng.integers(2, 41)))
        x = synthetic_frames(B, 8 * ((H + 7) // 8), 8 * ((W + 7) // 8), seed=100 + case)[:, :, :H, :W].contiguous()
        with torch.no_grad():
            zr = cpu_ref.revnet_forward(x, sd, sp)
            z = net(x.cuda())
            assert_close(z, zr, TIGHT, f"{mode} {B}x{H}x{W} forward")
            zp = zr + 0.02 * torch.randn(zr.shape, generator=torch.Generator().manual_seed(case))
            assert_close(net(zp.cuda(), forward=False), cpu_ref.revnet_inverse(zp, sd, sp), TIGHT, f"{mode} {B}x{H}x{W} inverse")


def test_mask_plan_equals_transfer():
    """plan_masks / bind_style / transfer_with_plan (masks and style fixed over a clip) == transfer(masks), bit for bit"""
    from models.cWCT import cWCT
    net, sd, sp = make_net("photo")
    cw = cWCT()
    B, H, W = 2, 48, 64
    cm = np.stack([synthetic_mask(H, W, 3, seed=1), synthetic_mask(H, W, 4, seed=2)])
    sm = np.stack([synthetic_mask(40, 56, 4, seed=3, speck=False)] * 2)
    with torch.no_grad():
        zs = net(synthetic_frames(1, 40, 56, seed=9).cuda()).expand(B, -1, -1, -1)
        plan = cw.plan_masks(cm, sm, (B, 32, H, W), zs.shape, zs.device)
        assert cw.plan_info(plan, 0) == ([0, 1, 2], False) and cw.plan_info(plan, 1) == ([0, 1, 2, 3], False)   # specks / missing labels dropped
        bound = cw.bind_style(cw.plan_masks(cm, sm, (B, 32, H, W), zs.shape, zs.device), zs)
        for seed in (4, 5):
            zc = net(synthetic_frames(B, H, W, seed=seed).cuda())
            ref = cw.transfer(zc.clone(), zs, cm, sm)
            assert torch.equal(cw.transfer_with_plan(zc.clone(), zs, plan), ref)
            assert torch.equal(cw.transfer_with_plan(zc.clone(), None, bound), ref)
        with pytest.raises(ValueError):
            cw.transfer_with_plan(zc[:, :, :32], zs, plan)
        with pytest.raises(ValueError):
            cw.transfer_with_plan(zc, None, plan)


def test_single_pass_masked_transfer_hard_cases():
    """the device-side label plan and the one-pass kernels on the cases per-label passes found easy to get wrong:
    per-pixel random labels (every tile holds every label), 11 labels (> 8 resident slots: two statistics passes),
    device-tensor masks, labels invalid by count or ratio, in place — all against the oracle"""
    from models.cWCT import cWCT
    cw = cWCT()
    rng = np.random.default_rng(12)
    H, W, sH, sW = 48, 80, 40, 64
    zc = torch.from_numpy(rng.standard_normal((1, 32, H, W)).astype(np.float32)) * 0.7 + 0.2
    zs = torch.from_numpy(rng.standard_normal((1, 32, sH, sW)).astype(np.float32)) * 1.3 - 0.1
    # labels 0..10 scattered per pixel over wide bands (>= 300 pixels each, so 32 channels are well conditioned); label 40 is a
    # 7-pixel speck, label 50 covers 600 content pixels but only 4 style pixels (ratio > 100 and count <= 10), label 60 exists
    # only in the style
    cm = (rng.integers(0, 11, (H, W))).astype(np.uint8)
    cm[0, :7] = 40
    cm[10:20, 10:70] = 50
    sm = (rng.integers(0, 11, (sH, sW))).astype(np.uint8)
    sm[0, :4] = 50
    sm[5:8, :] = 60
    ref = cpu_ref.transfer_seg(zc, zs, cm[None], sm[None])
    labels, ok = cpu_ref.compute_label_info(cm, sm)
    valid = [int(l) for l in labels if ok[l]]
    assert valid == list(range(11))
    plan = cw.plan_masks(cm[None], sm[None], zc.shape, zs.shape, torch.device("cuda"))
    assert cw.plan_info(plan) == (valid, False)
    out = cw.transfer(zc.cuda(), zs.cuda(), cm[None], sm[None])
    assert_close(out, ref, 2e-4, "11 scattered labels", tol_max=TOL)
    keep = T(np.isin(cm, [40, 50]))
    assert torch.equal(out[0][:, keep].cpu(), zc[0][:, keep])                       # invalid labels: content kept bit for bit
    # device-tensor masks, a learnt slot count, in place: identical results
    out_dev = cw.transfer(zc.cuda(), zs.cuda(), torch.from_numpy(cm)[None].cuda(), torch.from_numpy(sm)[None].cuda())
    assert torch.equal(out_dev, out)
    plan = cw.learn_slots(plan)
    assert plan.max_slots == 11
    zc_dev = zc.cuda()
    same = cw.transfer_with_plan(zc_dev, zs.cuda(), plan, inplace=True)
    assert same.data_ptr() == zc_dev.data_ptr() and torch.equal(same, out)
    # artistic codes: N = 128, two slots per statistics pass, one per apply pass
    zc128 = torch.from_numpy(rng.standard_normal((1, 128, 64, 64)).astype(np.float32))
    zs128 = torch.from_numpy(rng.standard_normal((1, 128, 64, 64)).astype(np.float32)) * 0.5 + 0.3
    cm3 = rng.integers(0, 3, (64, 64)).astype(np.uint8)
    sm3 = rng.integers(0, 3, (64, 64)).astype(np.uint8)
    ref128 = cpu_ref.transfer_seg(zc128, zs128, cm3[None], sm3[None])
    out128 = cw.transfer(zc128.cuda(), zs128.cuda(), cm3[None], sm3[None])
    assert_close(out128, ref128, 2e-4, "3 scattered labels, N = 128", tol_max=TOL)


def test_transfer_with_stats_inplace():
    from models.cWCT import cWCT
    net, sd, sp = make_net("photo")
    net.packed_code = False                     # the dense [B,32,H,W] form of the code (packed codes are never written by cWCT)
    cw = cWCT()
    with torch.no_grad():
        stats = cw.style_stats(net(synthetic_frames(1, 40, 56, seed=9).cuda()))
        zc = net(synthetic_frames(2, 48, 64, seed=3).cuda())
        ref = cw.transfer_with_stats(zc, stats)
        keep = zc.clone()
        out = cw.transfer_with_stats(zc, stats, inplace=True)
        assert out.data_ptr() == zc.data_ptr() and torch.equal(out, ref) and not torch.equal(zc, keep)


@pytest.mark.parametrize("precision", ["f16x2h", "f16x2", "bf16x3", "fp32"])
def test_packed_code_equals_dense_path(precision):
    """net(x) in photorealistic mode returns a PackedCode (vstnet_amd/code.py): the code in the coupling blocks' own layout.
    It must be the same [B,32,H,W] tensor to every caller, and encode -> cWCT -> decode on the packed rows must give what the
    dense path (spread, NCHW cWCT, gather; checked against the oracle elsewhere in this file) gives.  Shapes include
    H*W not a multiple of 32 (row tiles straddle the halves / the end) and a batch."""
    from models.cWCT import cWCT
    from vstnet_amd.code import PackedCode, from_dense
    net, sd, sp = make_net("photo", precision)
    net.packed_code = "always"                  # (by default batches of small images keep the dense, batched passes)
    dense, _, _ = make_net("photo", precision)
    dense.packed_code = False
    cw = cWCT(precision=precision)
    for B, H, W in ((1, 12, 20), (2, 40, 24), (1, 64, 96), (1, 256, 320)):
        x, xs = synthetic_frames(B, H, W, seed=5).cuda(), synthetic_frames(B, H + 8, W - 4, seed=6).cuda()
        with torch.no_grad():
            z, zd = net(x), dense(x)
            assert isinstance(z, PackedCode) and not isinstance(zd, PackedCode)
            assert z.shape == zd.shape and z.dtype == zd.dtype and z.device == zd.device
            assert torch.equal(z.materialize(), zd), "the packed rows are a permutation of z"
            assert torch.equal(z * 2.0, zd * 2.0) and torch.equal(z[:, 3:5].contiguous(), zd[:, 3:5].contiguous())
            assert torch.equal(from_dense(zd).materialize(), zd)
            assert torch.equal(net(z, forward=False), dense(zd, forward=False)), "plain decode"
            assert torch.equal(net(from_dense(zd), forward=False), dense(zd, forward=False))
            zs, zsd = net(xs), dense(xs)
            t, td = cw.transfer(z, zs), cw.transfer(zd, zsd)
            assert isinstance(t, PackedCode) and t.pending_affines is not None
            assert_close(t.materialize(), td, 2e-5, f"transfer on packed rows {B}x{H}x{W}")
            # (the two z_cs differ by ~1e-6; under f16x2h the rounded intermediates turn part of that into fp16 steps)
            tol_out = 1e-4 if precision == "f16x2h" else 2e-5
            out, outd = net(t, forward=False), dense(td, forward=False)
            assert float((out - outd).abs().max()) <= tol_out
            # cached style statistics, a mix of two styles with a content share, and the uint8 edge
            assert float((net(cw.transfer_with_stats(z, cw.style_stats(zs)), forward=False) - outd).abs().max()) <= tol_out
            mix, mixd = cw.interpolation(z, [zs, z], [0.6, 0.4], 0.3), cw.interpolation(zd, [zsd, zd], [0.6, 0.4], 0.3)
            assert_close(mix.materialize(), mixd, 2e-5, "interpolation on packed rows")
            u8, u8d = net.inverse_u8(t), dense.inverse_u8(td)
            assert int((u8.int() - u8d.int()).abs().max()) <= 1            # truncation of values 1e-6 apart
            # a second cWCT on a result, and a masked one, fall back to the dense code
            again = cw.transfer(t, zs)
            assert not isinstance(again, PackedCode)
            assert_close(again, cw.transfer(td, zsd), 2e-5, "cWCT of a cWCT result")
    auto, _, _ = make_net("photo", precision)
    with torch.no_grad():
        assert isinstance(auto(synthetic_frames(1, 16, 16).cuda()), PackedCode)
        assert not isinstance(auto(synthetic_frames(3, 16, 16).cuda()), PackedCode), "small-image batches stay dense"
    # parity with the oracle through the packed path
    x, xs = synthetic_frames(1, 48, 64, seed=0), synthetic_frames(1, 48, 64, seed=1)
    with torch.no_grad():
        got = net(cw.transfer(net(x.cuda()), net(xs.cuda())), forward=False)
    zc, zs_ = cpu_ref.revnet_forward(x, sd, sp), cpu_ref.revnet_forward(xs, sd, sp)
    ref = cpu_ref.revnet_inverse(cpu_ref.transfer(zc, zs_), sd, sp)
    assert_close(got, ref, NET_TOL[precision], "stylised frame through the packed code vs oracle")


@pytest.mark.parametrize("precision", ["bf16x3", "f16x2", "f16x2h"])
@pytest.mark.parametrize("kind", ["bands", "noise"])
def test_packed_code_masked_transfer(kind, precision):
    """transfer_with_plan on a PackedCode: per-label statistics and per-row affine maps on the packed rows (label map
    permuted once per plan into the rows' order), applied by the inverse pass.  Against the dense masked route, and against
    the oracle; 3..8 labels incl. a speck (invalid label: its rows keep their values), a batch, region and per-pixel masks."""
    from models.cWCT import cWCT
    from vstnet_amd.code import PackedCode
    # (under the fp16 modes the decode takes its own branch: the per-row maps write half 0 straight into split fp16 planes,
    # cwct_apply_labels_pm_kernel with planes0, and the inverse pass reads that half only through them)
    net, sd, sp = make_net("photo", precision)
    net.packed_code = "always"
    dense, _, _ = make_net("photo", precision)
    dense.packed_code = False
    tol_out = 1e-4 if precision == "bf16x3" else 3e-4      # fp16 intermediates turn 1e-6 differences of z_cs into fp16 steps
    cw = cWCT(precision="fp32")                  # dense masked apply in exact fp32 too: the two routes then differ by rounding only
    for B, H, W, K in ((1, 64, 96, 3), (2, 40, 24, 4), (1, 128, 192, 5), (1, 96, 96, 8)):
        x, xs = synthetic_frames(B, H, W, seed=2).cuda(), synthetic_frames(B, H, W, seed=3).cuda()
        cm = np.stack([synthetic_mask(H, W, K, seed=3 + b, kind=kind) for b in range(B)])
        sm = np.stack([synthetic_mask(H, W, K, seed=9 + b, speck=False, kind=kind) for b in range(B)])
        with torch.no_grad():
            z, zd, zs = net(x), dense(x), dense(xs)
            plan = cw.learn_slots(cw.plan_masks(cm, sm, z.shape, zs.shape, z.device))
            t, td = cw.transfer_with_plan(z, zs, plan), cw.transfer_with_plan(zd, zs, plan)
            assert isinstance(t, PackedCode) and t.pending_labels is not None and not isinstance(td, PackedCode)
            # (two summation orders of the per-label statistics; small regions have ill-conditioned covariances that amplify it)
            assert_close(t.materialize(), td, 1e-5, f"masked transfer on packed rows {kind} {B}x{H}x{W} K={K}", tol_max=1e-4)
            out_packed = net(t, forward=False)
            assert float((out_packed - dense(td, forward=False)).abs().max()) <= tol_out
            assert int((net.inverse_u8(t).int() - dense.inverse_u8(td).int()).abs().max()) <= 1
            bound = cw.bind_style(cw.learn_slots(cw.plan_masks(cm, sm, z.shape, zs.shape, z.device)), zs)
            assert_close(cw.transfer_with_plan(z, None, bound).materialize(), td, 1e-5, "bound style", tol_max=1e-4)
            ref = cpu_ref.transfer_seg(zd.cpu(), zs.cpu(), cm, sm)
            ref_out = cpu_ref.revnet_inverse(ref, sd, sp)
        assert_close(t.materialize(), ref, 2e-4, f"masked packed transfer vs oracle {kind} K={K}", tol_max=TOL)
        # and the decode of the pending-labels code against the ORACLE's inverse of the oracle's masked transfer
        assert_close(out_packed, ref_out, 3e-4, f"decode of the masked packed code vs oracle {kind} K={K} ({precision})", tol_max=TOL)
    # more than 8 slots, or a plan whose slot count was never read back: the dense route takes over
    H, W, K = 96, 128, 11
    x, xs = synthetic_frames(1, H, W, seed=2).cuda(), synthetic_frames(1, H, W, seed=3).cuda()
    cm = synthetic_mask(H, W, K, seed=5, kind=kind, speck=False)[None]
    sm = synthetic_mask(H, W, K, seed=6, kind=kind, speck=False)[None]
    with torch.no_grad():
        z, zs = net(x), dense(xs)
        plan = cw.plan_masks(cm, sm, z.shape, zs.shape, z.device)
        out0 = cw.transfer_with_plan(z, zs, plan)
        assert not isinstance(out0, PackedCode)
        out1 = cw.transfer_with_plan(z, zs, cw.learn_slots(plan))
        assert (not isinstance(out1, PackedCode)) == (plan.max_slots > 8)
        assert_close(out1 if not isinstance(out1, PackedCode) else out1.materialize(), out0, 1e-5, "11 labels", tol_max=1e-4)


@pytest.mark.parametrize("precision", ["f16x2h", "f16x2", "bf16x3"])
def test_packed_code_artistic(precision):
    """The artistic net's code [B,128,H/2,W/2] as a PackedCode: rows of 128 floats; statistics on ten 32 x 32 MFMA blocks,
    the affine map on the exact-fp32 MFMA inside the decode.  Against the dense route and the oracle."""
    from models.cWCT import cWCT
    from vstnet_amd.code import PackedCode, from_dense
    net, sd, sp = make_net("art", precision)
    net.packed_code = "always"
    dense, _, _ = make_net("art", precision)
    dense.packed_code = False
    cw = cWCT(precision="fp32")                  # the dense N = 128 apply in exact fp32 as well
    for B, H, W, tol in ((1, 64, 96, 1e-4), (2, 40, 24, 5e-4), (1, 256, 256, 2e-5), (1, 12, 20, 5e-4)):
        x, xs = synthetic_frames(B, H, W, seed=5).cuda(), synthetic_frames(B, H, W, seed=6).cuda()
        with torch.no_grad():
            z, zd = net(x), dense(x)
            assert isinstance(z, PackedCode) and tuple(z.shape) == tuple(zd.shape) == (B, 128, H // 2, W // 2)
            assert torch.equal(z.materialize(), zd) and torch.equal(from_dense(zd).materialize(), zd)
            assert torch.equal(net(z, forward=False), dense(zd, forward=False)), "plain decode"
            zs, zsd = net(xs), dense(xs)
            t, td = cw.transfer(z, zs), cw.transfer(zd, zsd)
            assert isinstance(t, PackedCode) and t.pending_affines is not None
            # two summation orders of a 128 x 128 covariance; with few pixels (L < 128: the jitter path) it is ill-conditioned
            assert_close(t.materialize(), td, tol, f"artistic transfer on packed rows {B}x{H}x{W}")
            assert float((net(t, forward=False) - dense(td, forward=False)).abs().max()) <= 5 * tol
            assert_close(net(cw.transfer_with_stats(z, cw.style_stats(zs)), forward=False), dense(td, forward=False), 5 * tol,
                         "cached style")
            assert int((net.inverse_u8(t).int() - dense.inverse_u8(td).int()).abs().max()) <= 1
    x, xs = synthetic_frames(1, 64, 64, seed=0), synthetic_frames(1, 64, 64, seed=1)
    with torch.no_grad():
        got = net(cw.transfer(net(x.cuda()), net(xs.cuda())), forward=False)
    zc, zs_ = cpu_ref.revnet_forward(x, sd, sp), cpu_ref.revnet_forward(xs, sd, sp)
    ref = cpu_ref.revnet_inverse(cpu_ref.transfer(zc, zs_), sd, sp)
    assert_close(got, ref, NET_TOL[precision], "artistic stylised frame through the packed code vs oracle")
