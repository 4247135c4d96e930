"""Can an MFMA-bound stage-3-like workgroup share a CU with the HBM-bound stage-1/2 kernels of another frame?

    hipcc -O3 --offload-arch=gfx950 -fPIC -shared -o tools/probe/libmfma_probe.so tools/probe/mfma_probe.hip
    python tools/overlap_probe.py

Stream A runs a synthetic MFMA-bound kernel (the LDS-read : MFMA ratio of conv_sp_kernel) with a given footprint
(waves per workgroup, LDS bytes; one workgroup per CU), stream B runs real coupling blocks of stage 1 or 2.  Prints each
alone and both together: together ~ max(alone) means the two kinds of work overlap on the CUs, ~ sum means they do not.
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstnet_amd import _lib                                     # noqa: E402
from vstnet_amd.synth import synthetic_state_dict               # noqa: E402
from models.RevResNet import RevResNet                          # noqa: E402


def timed(fn, streams):
    for s in streams:
        s.synchronize()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in streams:
        s.wait_event(e0)
    fn()
    for s in streams:
        e = torch.cuda.Event()
        e.record(s)
        torch.cuda.current_stream().wait_event(e)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    L = _lib.lib()
    P = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libmfma_probe.so"))
    P.mfma_probe.restype = C.c_int
    P.mfma_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    H = W = args.size
    dev = torch.device("cuda", 0)
    net = RevResNet().to(dev).eval()
    net.load_state_dict(synthetic_state_dict())
    w = net._ensure_packed(dev)
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    sink = torch.zeros(16, device=dev)
    kidx = {(16, 1): 3, (64, 1): 13}
    bufs = {}
    for ch, div in ((16, 1), (64, 2)):
        bufs[ch] = (torch.randn(1, H // div, W // div, ch, device=dev), torch.randn(1, H // div, W // div, ch, device=dev),
                    torch.empty(L.vst_block_tmp_bytes(1, H, W), dtype=torch.uint8, device=dev))

    def blocks(ch, n):
        dst, src, tmp = bufs[ch]
        st = C.c_void_p(sb.cuda_stream)
        for _ in range(n):
            _lib.check(L.vst_block_apply(C.byref(w.blocks[kidx[(ch, 1)]]), ch, 1, 1, _lib.PREC_F16X2, C.c_void_p(dst.data_ptr()),
                                         C.c_void_p(src.data_ptr()), C.c_void_p(tmp.data_ptr()), 1, H, W, st), "block")

    def mfma(waves, lds, iters, n, grid=256):
        for _ in range(n):
            rc = P.mfma_probe(waves, lds, iters, grid, C.c_void_p(sink.data_ptr()), C.c_void_p(sa.cuda_stream))
            assert rc == 0, rc

    n = args.reps
    for waves, lds, grid, iters in ((8, 156 * 1024, 256, 8), (4, 84 * 1024, 256, 16), (4, 84 * 1024, 512, 8), (4, 40 * 1024, 512, 8)):
        mfma(waves, lds, iters, 3, grid)
        t_m = timed(lambda: mfma(waves, lds, iters, n, grid), [sa])
        print(f"MFMA probe: {waves} waves, {lds // 1024} KB LDS, grid {grid}, {iters * 9 * 16} MFMAs per wave: {t_m / n:7.1f} us per launch alone", flush=True)
        for ch in (16, 64):
            blocks(ch, 3)
            t_b = timed(lambda: blocks(ch, n), [sb])

            def both():
                # interleave the enqueue so that neither stream's launches wait on the host
                dst, src, tmp = bufs[ch]
                for _ in range(n):
                    mfma(waves, lds, iters, 1, grid)
                    blocks(ch, 1)
            t_x = timed(both, [sa, sb])
            print(f"    + stage-{1 if ch == 16 else 2} blocks ({t_b / n:6.1f} us per block alone): together {t_x / n:7.1f} us per pair "
                  f"(sum {(t_m + t_b) / n:6.1f}, max {max(t_m, t_b) / n:6.1f})", flush=True)


if __name__ == "__main__":
    main()
