"""Per-CU residency of the conv kernels of several frames in flight (diagnostic -DVST_TRACE=3 build only).

    python tools/ab_build.py vstnet_amd/abl/trace3.so -DVST_TRACE=3
    VSTNET_HIP_LIB=$PWD/vstnet_amd/abl/trace3.so python tools/trace_cu.py --streams 3 --lean 1 [--frames 9]

Every workgroup of every conv launch leaves {start, end (100 MHz), HW_ID, XCC_ID, class, Cin, Cout}.  S3 = conv_pipe_kernel
(the MFMA-bound 256-channel convs), S12 = the HBM-bound 16- / 64-channel kernels.  Reported over the steady middle of the
run: how much of the CU-time had an S3 workgroup resident, an S12 workgroup resident, BOTH on the same CU at the same time
(the co-residency VST_OPT_STAGE3_LEAN is for), or neither; and the same for the chip as a whole (any CU).
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstnet_amd import _lib                                     # noqa: E402
from vstnet_amd.synth import synthetic_state_dict, synthetic_frames   # noqa: E402
from models.RevResNet import RevResNet                          # noqa: E402
from models.cWCT import cWCT                                    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=9)
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--lean", type=int, default=1)
    ap.add_argument("--out", default=None, help="write the summary JSON here")
    ap.add_argument("--pair", type=int, default=0, help="16 / 64: instead of frames, stream A runs 256-channel blocks and stream B "
                    "blocks of this channel count, back to back (the two-stream experiment of tools/overlap_real.py)")
    ap.add_argument("--dump-cu", type=int, default=-1, help="print the workgroup timeline of this CU (index into the sorted CU ids)")
    ap.add_argument("--dump-us", default="0.4,0.41", help="window of the dump as fractions of the span")
    args = ap.parse_args()
    L = _lib.lib()
    L.vst_trace_set.restype = C.c_int
    L.vst_trace_set.argtypes = [C.c_void_p, C.c_uint]
    L.vst_trace_count.restype = C.c_int
    L.vst_trace_count.argtypes = [C.POINTER(C.c_uint)]
    dev = torch.device("cuda", 0)
    net = RevResNet(precision="bf16x3")
    net.load_state_dict(synthetic_state_dict(1234))
    net = net.to(dev).eval()
    cw = cWCT(precision="bf16x3")
    _lib.set_option(_lib.OPT_STAGE3_LEAN, args.lean)
    cap = 400000 * args.frames
    buf = torch.zeros(cap, 4, dtype=torch.int64, device=dev)
    with torch.no_grad():
        S = args.size
        content = synthetic_frames(1, S, S, seed=0).to(dev)
        style = synthetic_frames(1, S, S, seed=1).to(dev)
        s_stats = cw.style_stats(net(style))
        streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)]

        def frame(i):
            with torch.cuda.stream(streams[i % args.streams]):
                z = net(content, forward=True)
                net(cw.transfer_with_stats(z, s_stats), forward=False)
        if args.pair:
            w = net._ensure_packed(dev)
            kidx = {16: 3, 64: 13, 256: 25}
            H = W = S
            bufs = {}
            for ch, div in ((16, 1), (64, 2), (256, 4)):
                bufs[ch] = (torch.randn(1, H // div, W // div, ch, device=dev), torch.randn(1, H // div, W // div, ch, device=dev),
                            torch.empty(L.vst_block_tmp_bytes(1, H, W), dtype=torch.uint8, device=dev))

            def blocks(ch, k, stream):
                dst, src, tmp = bufs[ch]
                for _ in range(k):
                    _lib.check(L.vst_block_apply(C.byref(w.blocks[kidx[ch]]), ch, 1, 1, _lib.PREC_BF16X3, C.c_void_p(dst.data_ptr()),
                                                 C.c_void_p(src.data_ptr()), C.c_void_p(tmp.data_ptr()), 1, H, W,
                                                 C.c_void_p(stream.cuda_stream)), "block")

            def frame(i):
                blocks(256, 1, streams[0])
                blocks(args.pair, 2, streams[1])
        for i in range(args.streams):
            frame(i)
        torch.cuda.synchronize()
        assert L.vst_trace_set(C.c_void_p(buf.data_ptr()), cap) == 0
        for i in range(args.frames):
            frame(i)
        torch.cuda.synchronize()
    n = C.c_uint(0)
    assert L.vst_trace_count(C.byref(n)) == 0
    n = min(n.value, cap)
    r = buf[:n].cpu().numpy().astype(np.int64)
    r = r[r[:, 1] > 0]                                           # (padding workgroups of a rounded-up grid leave nothing)
    n = len(r)
    t0, t1, hw, tag = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    base = t0.min()
    t0, t1 = (t0 - base) * 0.01, (t1 - base) * 0.01              # microseconds
    xcc = (hw >> 32) & 15
    cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    queue = ((hw >> 24) & 7) | (((hw >> 6) & 3) << 3) | (((hw >> 30) & 3) << 5)      # QUEUE_ID, PIPE_ID, ME_ID
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    klass = tag & 255
    s3 = klass == 4
    ids = np.unique(cuid)
    span = t1.max()
    lo, hi = 0.15 * span, 0.85 * span                            # steady middle: every stream busy
    grid = np.linspace(lo, hi, 3000, endpoint=False)
    live3 = np.zeros((len(ids), len(grid)), dtype=np.int16)
    live12 = np.zeros_like(live3)
    for k, i in enumerate(ids):
        m = cuid == i
        a, b, c3 = t0[m], t1[m], s3[m]
        live = (a[:, None] <= grid[None, :]) & (b[:, None] > grid[None, :])
        live3[k] = live[c3].sum(0)
        live12[k] = live[~c3].sum(0)
    has3, has12 = live3 > 0, live12 > 0
    rec = {
        "lean": args.lean, "streams": args.streams, "frames": args.frames, "workgroups": int(n), "cus": int(len(ids)),
        "queues_seen": sorted(int(q) for q in np.unique(queue)), "span_us": round(float(span), 1),
        "frames_per_s_traced": round(args.frames / (span * 1e-6), 1),
        "cu_time_share": {"s3_only": round(float(np.mean(has3 & ~has12)), 4), "s12_only": round(float(np.mean(~has3 & has12)), 4),
                          "both_on_same_cu": round(float(np.mean(has3 & has12)), 4), "idle": round(float(np.mean(~has3 & ~has12)), 4)},
        "chip_time_share": {"s3_somewhere_only": round(float(np.mean(has3.any(0) & ~has12.any(0))), 4),
                            "s12_somewhere_only": round(float(np.mean(~has3.any(0) & has12.any(0))), 4),
                            "both_somewhere": round(float(np.mean(has3.any(0) & has12.any(0))), 4)},
        "mean_live_wg_per_cu": {"s3": round(float(live3.mean()), 3), "s12": round(float(live12.mean()), 3)},
        "wg_life_us": {"s3_mean": round(float((t1 - t0)[s3].mean()), 2), "s12_mean": round(float((t1 - t0)[~s3].mean()), 2)},
    }
    if args.dump_cu >= 0:
        f0, f1 = (float(v) for v in args.dump_us.split(","))
        m = (cuid == ids[args.dump_cu]) & (t1 > f0 * span) & (t0 < f1 * span)
        order = np.argsort(t0[m])
        cin, cout = (tag >> 8) & 4095, (tag >> 20) & 4095
        for j in order:
            k = np.flatnonzero(m)[j]
            print(f"  {t0[k]:10.2f} -> {t1[k]:10.2f} ({t1[k] - t0[k]:6.2f} us)  q{queue[k]:3d}  class {klass[k]} <{cin[k]},{cout[k]}>  simd-wave {hw[k] & 63:2d}")
    print(json.dumps(rec, indent=1))
    if args.out:
        json.dump(rec, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
