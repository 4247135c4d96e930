"""Time one coupling block (vst_block_apply) or whole passes per precision mode on the GPU.

    python tools/bench_block.py [--size 1024] [--iters 50]

Prints microseconds per block for the 256-channel stride-1 block in every precision mode and the forward+inverse
pass time, so kernel variants can be compared without the rest of the frame.
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstnet_amd import _lib                                     # noqa: E402
from vstnet_amd.synth import synthetic_state_dict, synthetic_frames   # noqa: E402
from models.RevResNet import RevResNet                          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--modes", default="bf16x3,f16x2")
    ap.add_argument("--blocks", default="256:1")
    ap.add_argument("--pingpong", type=int, default=None, help="VST_OPT_STAGE3_PINGPONG for this run")
    ap.add_argument("--lean", type=int, default=None, help="VST_OPT_STAGE3_LEAN for this run")
    args = ap.parse_args()
    L = _lib.lib()
    if args.pingpong is not None:
        _lib.set_option(_lib.OPT_STAGE3_PINGPONG, args.pingpong)
    if args.lean is not None:
        _lib.set_option(_lib.OPT_STAGE3_LEAN, args.lean)
    H = W = args.size
    dev = torch.device("cuda", 0)
    modes = {"bf16x3": _lib.PREC_BF16X3, "f16x2": _lib.PREC_F16X2, "f16x2h": _lib.PREC_F16X2H}
    net = RevResNet().to(dev).eval()
    net.load_state_dict(synthetic_state_dict())
    w = net._ensure_packed(dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dst = torch.randn(1, H // 4, W // 4, 256, device=dev)
    src = torch.randn(1, H // 4, W // 4, 256, device=dev)
    tmp = torch.empty(L.vst_block_tmp_bytes(1, H, W), dtype=torch.uint8, device=dev)
    kidx = {(16, 1): 3, (64, 1): 13, (64, 2): 10, (256, 1): 25, (256, 2): 20}
    for spec in args.blocks.split(","):
        ch, stride = (int(v) for v in spec.split(":"))
        for name in args.modes.split(","):
            prec = modes[name]

            def run():
                _lib.check(L.vst_block_apply(C.byref(w.blocks[kidx[(ch, stride)]]), ch, stride, 1, prec,
                                             C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()),
                                             C.c_void_p(tmp.data_ptr()), 1, H, W, st), "block")
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            line = f"block c{ch}s{stride} {name:11s}: {e0.elapsed_time(e1) / args.iters * 1e3:8.1f} us"
            if ch == 256 and stride == 1:     # per-conv HIP-event times through the library's profile hook
                for cin, cout in ((256, 64), (64, 64), (64, 256)):
                    _lib.check(L.vst_profile_begin(_lib.kernel_id(cin, cout, 1), args.iters), "profile_begin")
                    for _ in range(args.iters):
                        run()
                    tot, n = C.c_double(), C.c_int()
                    _lib.check(L.vst_profile_end(C.byref(tot), C.byref(n)), "profile_end")
                    line += f"   {cin}->{cout}: {tot.value / max(n.value, 1) * 1e3:6.1f}"
            if os.environ.get("VST_STAMPS"):      # VST_SP_ABLATE & 8 builds: conv.4's workgroup stamps sit at the head of h2
                mid = (H // 4) * (W // 4) * 64 * 4
                st_ = tmp[mid:mid + 16].view(torch.int64).cpu()
                line += f"   conv.4 WG: {int(st_[0])} cyc / {int(st_[1]) * 10} ns = {int(st_[0]) / max(int(st_[1]), 1) * 0.1:.2f} GHz"
                st_ = tmp[0:16].view(torch.int64).cpu()
                line += f"   conv.1 WG: {int(st_[0])} cyc / {int(st_[1]) * 10} ns = {int(st_[0]) / max(int(st_[1]), 1) * 0.1:.2f} GHz"
                for nm, buf in (("conv.1", tmp[0:528]), ("conv.7", dst.reshape(-1)[:132].view(torch.uint8))):
                    v = buf.view(torch.int64).cpu().tolist()
                    line += f"\n      {nm} total {v[0]} cyc; per stage [loads issued, k=2, k=5, mfma done, dma landed, epilogue done, barrier passed]: " + \
                            " | ".join(" ".join(str(x) for x in v[2 + 7 * i: 9 + 7 * i]) for i in range(8))
            print(line, flush=True)
    x = synthetic_frames(1, H, W).to(dev)
    for name in args.modes.split(","):
        net.precision = net.resolved_precision = name
        z = net(x)
        for _ in range(3):
            net(net(x), forward=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = net(net(x), forward=False)
        e1.record()
        torch.cuda.synchronize()
        err = float((y - x).abs().max())
        print(f"fwd+inv {name:11s}: {e0.elapsed_time(e1) / 10:8.3f} ms   max|inv(fwd(x)) - x| = {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
