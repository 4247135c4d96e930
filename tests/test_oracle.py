"""CPU: the oracle restatement (oracle/cpu_ref.py) against the golden vectors minted from the real
reference by oracle/make_golden.py.  Runs anywhere (no GPU, no /root/reference)."""
import numpy as np
import torch

from oracle import cpu_ref
from vstnet_amd.synth import synthetic_state_dict, synthetic_frames

T = torch.from_numpy


def close(a, b, tol=2e-5):
    a, b = a.double(), b.double()
    assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))


def test_glue(golden):
    g = golden("glue")
    x = T(g["x"])
    assert torch.equal(cpu_ref.squeeze(x), T(g["squeeze"]))
    assert torch.equal(cpu_ref.unsqueeze(T(g["squeeze"])), T(g["unsqueeze_of_squeeze"]))
    assert torch.equal(cpu_ref.unsqueeze(cpu_ref.squeeze(x)), x)
    assert torch.equal(cpu_ref.inj_pad_fwd(x, 5), T(g["inj_pad5"]))
    assert torch.equal(cpu_ref.inj_pad_inv(T(g["inj_pad5"]), 5), x)


BLOCKS = [("c16s1", "stack.3.", 1), ("c64s2", "stack.10.", 2), ("c64s1", "stack.13.", 1),
          ("c256s2", "stack.20.", 2), ("c256s1", "stack.25.", 1), ("cr0", "channel_reduction.block_list.0.", 1)]


def test_weights_are_the_fixture_weights(golden):
    g = golden("blocks")
    sd = synthetic_state_dict(int(g["weights_seed"]))
    assert abs(sum(float(v.double().sum()) for v in sd.values()) - float(g["weights_checksum"])) < 1e-9
    assert len(sd) == 192 and sum(v.numel() for v in sd.values()) == 4089936


def test_blocks(golden):
    g = golden("blocks")
    sd = synthetic_state_dict(int(g["weights_seed"]))
    for nm, prefix, stride in BLOCKS:
        x1, x2 = T(g[f"{nm}_x1"]), T(g[f"{nm}_x2"])
        o2, y1 = cpu_ref.block_forward(x1, x2, sd, prefix, stride)
        close(o2, T(g[f"{nm}_out_x2"]), 0)
        close(y1, T(g[f"{nm}_out_y1"]))
        i1, i2 = cpu_ref.block_inverse(T(g[f"{nm}_out_x2"]), T(g[f"{nm}_out_y1"]), sd, prefix, stride)
        close(i1, T(g[f"{nm}_inv_x1"]))
        close(i2, T(g[f"{nm}_inv_x2"]), 0)


def test_network(golden):
    for mode, hd, sp in (("photo", 16, 2), ("art", 64, 1)):
        g = golden(f"net_{mode}")
        sd = synthetic_state_dict(int(g["weights_seed"]), hd, sp)
        for tag in ("16", "24x40", "32b2"):
            x = T(g[f"x_{tag}"])
            b, _, h, w = x.shape
            assert torch.equal(x, synthetic_frames(b, h, w, seed=int(g["frames_seed"])))
            with torch.no_grad():
                close(cpu_ref.revnet_forward(x, sd, sp), T(g[f"z_{tag}"]))
                close(cpu_ref.revnet_inverse(T(g[f"zp_{tag}"]), sd, sp), T(g[f"y_{tag}"]))
                close(cpu_ref.revnet_inverse(T(g[f"z_{tag}"]), sd, sp), x, 5e-6)


def test_cwct_2d(golden):
    g = golden("cwct_2d")
    for N, L in ((32, 50), (32, 4096), (128, 1024)):
        c, s = T(g[f"c_{N}_{L}"]), T(g[f"s_{N}_{L}"])
        wh = cpu_ref.whitening(c)
        close(wh, T(g[f"whiten_{N}_{L}"]), 2e-4)
        close(cpu_ref.coloring(T(g[f"whiten_{N}_{L}"]), s), T(g[f"color_{N}_{L}"]))
        # property: whitened features have identity covariance; coloured ones the style's covariance
        cov = lambda m: (m - m.mean(-1, keepdim=True)) @ (m - m.mean(-1, keepdim=True)).t() / (m.shape[1] - 1)
        if L >= 8 * N:       # well-conditioned cases only (L=50 has cond ~1e6)
            assert float((cov(wh.double()) - torch.eye(N)).abs().max()) < 2e-2
            out = T(g[f"color_{N}_{L}"]).double()
            assert float((cov(out) - cov(s.double())).abs().max()) < 2e-2 * float(cov(s.double()).abs().max())


def test_cwct_transfer_and_interpolation(golden):
    g = golden("cwct_transfer")
    c, s, sb = T(g["c"]), T(g["s"]), T(g["s_b"])
    assert bool(g["fork_transfer_raises"])          # documents SURVEY 8(a) C-1
    close(cpu_ref.transfer(c, s), T(g["transfer"]))
    close(cpu_ref.interpolation(c, [s], [1.0], 0.0), T(g["interp1_ac0"]))
    close(cpu_ref.interpolation(c, [s], [1.0], 0.3), T(g["interp1_ac03"]))
    close(cpu_ref.interpolation(c, [s, sb], [0.6, 0.4], 0.0), T(g["interp2_ac0.0"]))
    close(cpu_ref.interpolation(c, [s, sb], [0.6, 0.4], 0.3), T(g["interp2_ac0.3"]))


def test_cwct_masked(golden):
    g = golden("cwct_masked")
    labels, ok = cpu_ref.compute_label_info(g["cmask"][0], g["smask"][0])
    assert list(labels) == list(g["labels"]) and np.array_equal(ok, g["valid"])
    out = cpu_ref.transfer_seg(T(g["c"]), T(g["s"]), g["cmask"], g["smask"])
    close(out, T(g["out"]), 5e-5)
    # invalid labels (speck, label missing from style) keep the content feature
    keep = np.isin(g["cmask"][0], [4, 9])
    assert torch.equal(out[0][:, T(keep)], T(g["c"])[0][:, T(keep)])


def test_cholesky_jitter(golden):
    g = golden("cwct_jitter")
    L, tries = cpu_ref.cholesky_dec(T(g["conv"]), return_tries=True)
    assert tries == int(g["tries"]) >= 1
    close(L, T(g["L"]), 1e-3)
    L1, t1 = cpu_ref.cholesky_dec(torch.ones(4, 4), return_tries=True)
    assert t1 == int(g["ones4_tries"]) == 1
    close(L1, T(g["ones4_L"]), 1e-4)
    L2, t2 = cpu_ref.cholesky_dec(T(g["neg_in"]), return_tries=True)
    assert t2 == int(g["neg_tries"]) == 2          # cumulative schedule: 2e-5, then +4e-5
    close(L2, T(g["neg_L"]), 1e-5)


def test_batch_coupled_jitter(golden):
    """[B,N,N] stacks are factored at once: one failing sample jitters the whole batch (models/cWCT.py:122-128)"""
    g = golden("cwct_batch_jitter")
    c, s = T(g["c"]), T(g["s"])
    for ac in (0.0, 0.3):
        close(cpu_ref.interpolation(c, [s], [1.0], ac), T(g[f"out_ac{ac}"]), 2e-5)
    close(cpu_ref.transfer(c, s), T(g["out_ac0.0"]), 2e-5)
    cc = c.reshape(2, 32, -1) - c.reshape(2, 32, -1).mean(-1, keepdim=True)
    _, tries = cpu_ref.cholesky_dec(cc @ cc.transpose(-1, -2) / 63, return_tries=True)
    assert tries == int(g["tries"]) == 1


def test_segremap_golden(golden):
    """mask producers (8(f) rank 3) against outputs of the reference's own SegReMapping on its real ADE20K relation table:
    the drop-in (histogram + LUT) and the oracle's loop restatement"""
    from models.segmentation.SegReMapping import SegReMapping
    g = golden("segremap")
    table = g["mapping"].astype(np.int64)
    assert table.shape == (150, 150) and np.array_equal(table[-1], np.arange(150))
    for impl in (SegReMapping(table, float(g["min_ratio"])), cpu_ref.SegReMappingLoop(table, float(g["min_ratio"]))):
        for t in range(int(g["n_cases"])):
            a = impl.self_remapping(g[f"seg_{t}"])
            b = impl.self_remapping(g[f"sty_{t}"])
            assert a.dtype == np.uint8 and np.array_equal(a, g[f"self_seg_{t}"]) and np.array_equal(b, g[f"self_sty_{t}"])
            assert np.array_equal(impl.cross_remapping(a, b), g[f"cross_{t}"])


def test_config1(golden):
    g = golden("config1_photo256")
    sd = synthetic_state_dict(int(g["weights_seed"]))
    xc = synthetic_frames(1, 256, 256, seed=int(g["content_seed"]))
    xs = synthetic_frames(1, 256, 256, seed=int(g["style_seed"]))
    with torch.no_grad():
        zc, zs, zcs, sty = cpu_ref.stylize(xc, xs, sd, 2)
    close(zc[:, :, ::8, ::8], T(g["zc_sub"]))
    close(zcs[:, :, ::8, ::8], T(g["zcs_sub"]), 5e-5)
    close(sty, T(g["stylized"]), 5e-5)
    diff = (cpu_ref.to_uint8(sty).int() - T(g["stylized_u8"]).int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 1e-3


def test_lab_luminance(golden):
    """the fork's Lab post-process (project/image_style/color.py, vstnet.py:189-220): bit-identical restatement"""
    g = golden("lab")
    c, s = T(g["content"]), T(g["stylized"])
    assert torch.equal(cpu_ref._rgb2lab(c), T(g["lab_content"]))
    out = cpu_ref.luminance_transfer(c, s)
    assert torch.equal(out, T(g["out"]))
    # properties: idempotent on its own output's luminance; identity when stylized == content (up to the Lab clamp)
    same = cpu_ref.luminance_transfer(c, c)
    assert float((same - c).abs().max()) < 2e-3
    gray = c.mean(1, keepdim=True).expand(-1, 3, -1, -1)
    lab = cpu_ref._rgb2lab(gray)
    assert float(lab[:, 1:].abs().max()) < 2e-3                       # greys have no chroma


def test_cwct_use_double(golden):
    """oracle with use_double=True against the goldens minted from the reference's cWCT(use_double=True)
    (models/cWCT.py:13-16,35-47,66,106,220,238,259) on an ill-conditioned content code (cond(cov) ~ 1e6)"""
    g = golden("cwct_double")
    c, s1, s2 = T(g["c"]), T(g["s1"]), T(g["s2"])
    for ac in (0.0, 0.3):
        out = cpu_ref.interpolation(c, [s1, s2], [0.7, 0.3], ac, use_double=True)
        assert out.dtype == torch.float32
        close(out, T(g[f"interp_ac{ac}"]), 1e-6)
        # the case separates the arithmetics: the oracle's fp32 path is ~1e-2 away from the fp64 golden
        d32 = (cpu_ref.interpolation(c, [s1, s2], [0.7, 0.3], ac) - T(g[f"interp_ac{ac}"])).abs().max()
        assert float(d32) > 1e-3
    close(cpu_ref.transfer_seg(c, s1, g["cmask"], g["smask"], use_double=True), T(g["masked"]), 1e-6)
    # the jitter branch under use_double (a constant channel: an exactly zero pivot -> one retry); the reference adds a float32
    # identity times eps to the float64 covariance (models/cWCT.py:120-124), and so does the oracle
    cj, sj = T(g["c_jit"]), T(g["s_jit"])
    out = cpu_ref.interpolation(cj, [sj], [1.0], 0.0, use_double=True)
    close(out, T(g["interp_jit"]), 1e-6)
    x = cj.reshape(1, 32, -1).double()
    x = x - x.mean(-1, keepdim=True)
    assert cpu_ref.cholesky_dec(x @ x.transpose(-1, -2) / 63, return_tries=True)[1] == int(g["jit_tries"]) == 1


def _general_case(g, tag):
    import ast
    arch = ast.literal_eval(str(g[f"{tag}_arch"]))
    sd = {k[len(tag) + 3:]: T(g[k]) for k in g.files if k.startswith(f"{tag}_w_")}
    return arch, sd


def test_general_architectures(golden):
    """the reference's other constructor arguments (models/RevResNet.py:166-201: nBlocks / nStrides / nChannels / in_channel / mult
    / hidden_dim / sp_steps / kernel), goldens minted from the reference for two such nets (weights inside the fixture)"""
    g = golden("net_general")
    for tag in ("A", "B", "C"):
        arch, sd = _general_case(g, tag)
        x = T(g[f"{tag}_x"])
        with torch.no_grad():
            close(cpu_ref.revnet_forward(x, sd, arch["sp_steps"], arch), T(g[f"{tag}_z"]))
            close(cpu_ref.revnet_inverse(T(g[f"{tag}_zp"]), sd, arch["sp_steps"], arch["in_channel"], arch), T(g[f"{tag}_y"]))
            close(cpu_ref.revnet_inverse(T(g[f"{tag}_z"]), sd, arch["sp_steps"], arch["in_channel"], arch), x, 5e-6)
