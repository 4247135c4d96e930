"""CPU: host logic — the C-ABI library loads and exports everything include/vstnet.h declares, the
drop-in classes keep the reference's state_dict / attribute contract, argument errors are loud, and
there is no CPU fallback.  No compute call is made (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import vstnet_amd
from vstnet_amd import _lib
from vstnet_amd.synth import synthetic_state_dict, state_dict_spec

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(REPO, "include", "vstnet.h")).read()
    declared = set(re.findall(r"\b(vst_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"vst_conv_weights", "vst_block_weights", "vst_net_weights"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_library_exports_only_the_header(lib):
    """ABI hygiene: the shared library's exported FUNCTIONS are exactly the header's declarations — no
    internal helper (vst3_*, vst_pack_input_k, ...) and none of the diagnostic scaffolding (vst_trace_dump of -DVST_TRACE
    builds) — and build() passes no -DVST_* flag, so the ablation / trace branches are compiled out of what ships."""
    import inspect
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    funcs = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "TtWw"}
    funcs = {f for f in funcs if not f.startswith("_")}                   # (_init / _fini)
    assert funcs == set(_lib.EXPORTS), funcs ^ set(_lib.EXPORTS)
    assert "vst_trace_dump" not in out and "vst3_" not in " ".join(funcs)
    src = inspect.getsource(_lib.build)
    assert "-DVST" not in src and "-D" not in src.replace("-DVST", "")


def test_sizes_and_errors_without_gpu(lib):
    assert lib.vst_version() >= 101
    # two state halves (64 B/px each) + h1/h2 (32 B/px) + the split planes of both halves for the F16X2 kernels (2 x 64 B/px)
    assert lib.vst_pass_workspace_bytes(1, 1024, 1024) == 1024 * 1024 * 288
    assert lib.vst_block_tmp_bytes(2, 64, 32) == 2 * 64 * 32 * 160
    # packed conv = fp32 taps-major + 2 x bf16 fragment sections (+ 1 fp16 section in the LDS-DMA kernels' K order for the
    # 256-channel blocks' shapes)
    # fp32 taps + bf16 hi + bf16 lo + conv3's permuted fp16 (stage-3 shapes) + fp16 in the bf16 order (2-term kernels)
    assert lib.vst_conv_packed_bytes(64, 256) == 9 * 256 * 64 * 4 + 4 * (72 * 4 * 64 * 16)
    assert lib.vst_conv_packed_bytes(64, 16) == 9 * 16 * 64 * 4 + 3 * (5 * 4 * 64 * 16)
    assert lib.vst_conv_packed_bytes(4, 16) == ((9 * 16 * 4 * 4 + 255) // 256 * 256) + 3 * (5 * 4 * 16 * 16)
    assert lib.vst_cwct_stats_workspace_bytes(32, 1 << 20) == 512 * (32 * 32 + 64 + 4) * 4
    # argument validation happens before any launch
    null = C.c_void_p(0)
    one = C.c_void_p(1)
    assert lib.vst_pack_input(null, one, one, 1, 3, 16, 16, null) == -1
    assert lib.vst_pack_input(one, one, one, 1, 3, 18, 16, null) == -2
    assert lib.vst_pack_input(one, one, one, 1, 3, 4, 16, null) == -2
    assert lib.vst_spread(one, one, one, 1, 16, 16, 3, null) == -3
    assert lib.vst_cwct_apply(one, one, 48, 100, one, null, 0, null) == -2
    assert lib.vst_cwct_stats(one, 32, 100, null, 0, one, null, null) == -4
    assert lib.vst_cwct_stats_f64(one, 32, 100, null, 0, one, null, null) == -4
    assert lib.vst_cwct_stats_f64_workspace_bytes(32, 4096) == 2 * (33 + 1024) * 8
    assert lib.vst_cwct_factor_f64_workspace_bytes(128) == 4 * 128 * 128 * 8
    assert lib.vst_cwct_apply_f64(one, one, 48, 100, one, null, 0, null) == -2
    assert lib.vst_normalize_block(one, one, one, one, one, 16, 65, 16, null, null) == -2
    assert lib.vst_range_flags(None, 0) == -1
    net = _lib.NetWeights()
    assert lib.vst_revnet_forward(C.byref(net), one, one, null, 1, 3, 16, 16, 2, 0, null) == -4
    assert lib.vst_revnet_forward(C.byref(net), one, one, one, 1, 3, 16, 16, 5, 0, null) == -3
    assert b"multiples of 4" in lib.vst_error_string(-2)


def test_state_dict_contract():
    from models.RevResNet import RevResNet     # the reference's import path
    for hd, sp in ((16, 2), (64, 1)):
        net = RevResNet(hidden_dim=hd, sp_steps=sp)
        sd = net.state_dict()
        spec = state_dict_spec(hd, sp)
        assert list(sd.keys()) == [k for k, _ in spec] and len(sd) == 192
        assert all(tuple(sd[k].shape) == tuple(s) for k, s in spec)
        assert sum(p.numel() for p in net.parameters()) == 4089936
        assert int(net.down_scale) == 4
        net.load_state_dict(synthetic_state_dict(1234, hd, sp))     # strict
        assert all(float(v.abs().sum()) > 0 for k, v in net.state_dict().items() if k.endswith("bias"))


def test_general_constructor_arguments(golden):
    """every argument of the reference's constructor (models/RevResNet.py:166-201) builds the reference's module tree: same
    state_dict keys and shapes as the goldens' weights (minted from the reference for two non-default architectures); the
    published architecture takes the tuned path, everything else the generic HIP ops; impossible stage sequences are refused"""
    import ast
    from models.RevResNet import RevResNet
    g = golden("net_general")
    for tag in ("A", "B"):
        arch = ast.literal_eval(str(g[f"{tag}_arch"]))
        net = RevResNet(**arch)
        assert net._generic and int(net.down_scale) == int(np.prod(arch["nStrides"]))
        want = {k[len(tag) + 3:]: g[k].shape for k in g.files if k.startswith(f"{tag}_w_")}
        have = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        assert have == {k: tuple(v) for k, v in want.items()}
        net.load_state_dict({k: torch.from_numpy(g[f"{tag}_w_{k}"]) for k in want})      # strict
    assert not RevResNet()._generic and not RevResNet(hidden_dim=64, sp_steps=1)._generic
    assert RevResNet(hidden_dim=32, sp_steps=2)._generic            # 512-channel channel_reduction: the generic path
    assert RevResNet(nBlocks=[4, 4, 4])._generic and len(RevResNet(nBlocks=[4, 4, 4]).stack) == 12
    assert RevResNet(nChannels=None, in_channel=4, nBlocks=[1, 1, 1], hidden_dim=8).in_ch == 8      # :179-180
    with pytest.raises(ValueError):
        RevResNet(nChannels=[16, 32, 128])                          # a stride-2 stage multiplies the channels by 4
    with pytest.raises(ValueError):
        RevResNet(nBlocks=[1, 1], nStrides=[1, 2, 2])
    with pytest.raises(NotImplementedError):
        RevResNet(kernel=4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        RevResNet(nBlocks=[1, 1, 1])(torch.zeros(1, 3, 16, 16))


def test_no_cpu_fallback():
    from models.RevResNet import RevResNet
    from models.cWCT import cWCT
    net = RevResNet()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 3, 16, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 32, 16, 16), forward=False)
    cw = cWCT()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cw.transfer(torch.zeros(1, 32, 8, 8), torch.zeros(1, 32, 8, 8))
    with pytest.raises(AssertionError):
        cw.interpolation(torch.zeros(1, 32, 8, 8), [torch.zeros(1, 32, 8, 8)], [0.5, 0.5])


def test_packed_code_host_side():
    """PackedCode (vstnet_amd/code.py) without a GPU: it presents [B,32,H,W] float32 metadata, carries its pending affine
    maps, and any torch operation on it goes to the HIP library (no torch / CPU fallback): here that must fail loudly."""
    from vstnet_amd.code import PackedCode
    rows = torch.zeros(2, 32 * 16 * 24)
    z = PackedCode(rows, 16, 24)
    assert tuple(z.shape) == (2, 32, 16, 24) and z.dtype == torch.float32 and z.device == rows.device
    assert z.pending_affines is None and z.packed is rows and "PackedCode" in repr(z)
    t = z.with_affines(torch.zeros(2, 32 * 32 + 32))
    assert t.packed is rows and t.pending_affines.shape == (2, 1056) and tuple(t.shape) == tuple(z.shape)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        z + 1.0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        t.materialize()


def test_packed_code_notices_writes():
    """in-place edits of a PackedCode (direct, through a view, out=) mark the packed rows stale; reads do not.
    (Host logic only: materialize() is replaced by a CPU stand-in; the GPU side is test_packed_code_inplace_edits_are_not_lost.)"""
    from vstnet_amd.code import PackedCode

    class P(PackedCode):
        def materialize(self):
            if self._dense is None:
                self._dense = torch.arange(2 * 32 * 16, dtype=torch.float32).reshape(2, 32, 4, 4).clone()
            return self._dense
    rows = torch.zeros(2, 32 * 16)
    with torch.no_grad():
        z = P(rows, 4, 4)
        _ = z + 1.0, z[:, 3:5].sum(), z.reshape(2, -1).mean()
        assert not z.stale
        z.mul_(2.0)
        assert z.stale and float(z.materialize()[0, 0, 0, 1]) == 2.0
        z = P(rows, 4, 4)
        z[:, :, 1:3] = 0.5
        assert z.stale and float(z.materialize()[0, 0, 1, 0]) == 0.5
        z = P(rows, 4, 4)
        v = z[:, 1]
        assert not z.stale
        v.add_(1.0)
        assert z.stale
        z = P(rows, 4, 4)
        torch.add(z, 1.0, out=z)
        assert z.stale
        assert not P(rows, 4, 4).with_affines(torch.zeros(2, 1056)).stale


def test_cwct_route_table_is_total_and_consistent():
    """the cWCT routes (layout x mask x N x slots x use_double) are chosen in ONE function; enumerate it.
    Every combination maps to a documented route, and the structural rules hold (the GPU side — each route against the
    oracle, with `last_route` asserted — is tests/test_gpu_parity.py::test_cwct_every_route_vs_oracle)."""
    import itertools
    from models.cWCT import cWCT
    seen = set()
    for packed, masked, N, sp, slots, dbl in itertools.product((False, True), (False, True), (16, 32, 64, 128), (1, 2),
                                                               (0, 1, 8, 9, 32), (False, True)):
        r = cWCT.route(packed, masked, N, sp, slots, dbl)
        assert r in cWCT.ROUTES
        seen.add(r)
        if dbl:
            assert r.endswith("_f64")                                  # fp64 never touches the packed rows
        if "packed" in r:
            assert packed and not dbl and ((N, sp) in ((32, 2), (128, 1)))
        if r == "masked_packed_rows":
            assert masked and N == 32 and sp == 2 and 1 <= slots <= 8
        if masked and not dbl and N == 16:
            assert r == "masked_per_label"
        assert masked == r.startswith("masked")
    assert seen == set(cWCT.ROUTES)                                    # no dead entry in the table
    with pytest.raises(NotImplementedError):
        cWCT.route(False, False, 48)


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "vstnet_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f
    for f in ("models/RevResNet.py", "models/cWCT.py"):
        assert "oracle" not in open(os.path.join(REPO, f)).read()


def test_compute_label_info_matches_oracle(golden):
    from oracle import cpu_ref
    from models.cWCT import cWCT
    g = golden("cwct_masked")
    labels, ok = cWCT().compute_label_info(g["cmask"][0], g["smask"][0])
    l2, ok2 = cpu_ref.compute_label_info(g["cmask"][0], g["smask"][0])
    assert list(labels) == list(l2) == list(g["labels"])
    assert np.array_equal(ok, ok2) and np.array_equal(ok, g["valid"])


def test_pipeline_host_helpers():
    """vstnet_amd.pipeline: the background frame source and the background sink (pure host logic; the GPU part is
    covered by test_frame_pipeline_matches_sequential)."""
    import time
    import numpy as np
    import pytest
    from vstnet_amd.pipeline import prefetch, AsyncSink, FramePipeline

    assert list(prefetch(iter(range(10)), ahead=3)) == list(range(10))
    assert list(prefetch(iter([]), ahead=1)) == []

    def bad_source():
        yield 1
        raise ValueError("decode failed")
    it = prefetch(bad_source(), ahead=2)
    assert next(it) == 1
    with pytest.raises(ValueError, match="decode failed"):
        next(it)

    got = []
    sink = AsyncSink(lambda i, a: (time.sleep(0.001), got.append((i, int(a.sum()))))[1], ahead=2)
    buf = np.zeros((4, 4, 3), dtype=np.uint8)
    for i in range(6):
        buf[:] = i                      # the sink must have copied the slot before it is overwritten
        sink(i, buf)
    sink.close()
    assert got == [(i, i * 48) for i in range(6)]

    def bad_write(i, a):
        raise OSError("disk full")
    sink = AsyncSink(bad_write)
    sink(0, buf)
    with pytest.raises(OSError, match="disk full"):
        sink.close()

    if not __import__("torch").cuda.is_available():
        with pytest.raises(RuntimeError, match="needs the GPU"):
            FramePipeline(None, None, 64, 64)


def test_prefetch_producer_stops_with_the_consumer():
    """a consumer that stops early (exception / break) must not leave the decode thread blocked on a full queue"""
    import threading
    import time
    from vstnet_amd.pipeline import prefetch
    produced = []

    def source():
        for i in range(1000):
            produced.append(i)
            yield i

    before = threading.active_count()
    gen = prefetch(source(), ahead=2)
    assert next(gen) == 0
    gen.close()                                           # consumer goes away with the queue full
    deadline = time.time() + 5
    while threading.active_count() > before and time.time() < deadline:
        time.sleep(0.05)
    assert threading.active_count() <= before and len(produced) < 20
    with pytest.raises(ZeroDivisionError):                # producer errors still reach the consumer
        list(prefetch((1 // (3 - i) for i in range(5)), ahead=2))


def test_packed_weights_follow_parameter_edits():
    """the packed-weight cache is keyed on the parameters' versions: a parent module's load_state_dict or an in-place edit
    repacks (no GPU here: only the cache key is checked)"""
    import torch
    from models.RevResNet import RevResNet
    net = RevResNet()
    convs = [blk.conv[ci] for blk in net._blocks() for ci in (1, 4, 7)]
    v0 = net._param_versions(convs)
    with torch.no_grad():
        net.stack[3].conv[4].bias.add_(1.0)
    v1 = net._param_versions(convs)
    assert v0 != v1
    holder = torch.nn.Sequential(net)
    holder.load_state_dict(holder.state_dict())           # recurses without calling net.load_state_dict
    assert net._param_versions(convs) != v1 and net._packed is None
    # writes through `.data` are invisible to the key (p.data is another tensor object with its own version counter, same
    # storage): the documented remedy is refresh_weights()
    v2 = net._param_versions(convs)
    net._packed = ("sentinel",)
    net.stack[0].conv[1].weight.data.copy_(torch.zeros_like(net.stack[0].conv[1].weight))
    assert net._param_versions(convs) == v2 and net._packed is not None
    net.refresh_weights()
    assert net._packed is None
