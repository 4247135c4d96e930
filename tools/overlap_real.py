"""Do the REAL stage-3 kernels (bf16x3, 8-wave or lean 4-wave form) overlap with stage-1 / stage-2 blocks of another stream?

    python tools/overlap_real.py [--size 1024] [--reps 20]

Stream A runs 256-channel coupling blocks (VST_OPT_STAGE3_LEAN = 0 / 1), stream B stage-1 or stage-2 blocks; each alone,
then both together.  together ~ max(alone): overlap; ~ sum: none.
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstnet_amd import _lib                                     # noqa: E402
from vstnet_amd.synth import synthetic_state_dict               # noqa: E402
from models.RevResNet import RevResNet                          # noqa: E402
from overlap_probe import timed                                 # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--prio", type=int, default=0, help="-1: the stage-3 stream is a high-priority stream")
    args = ap.parse_args()
    L = _lib.lib()
    H = W = args.size
    dev = torch.device("cuda", 0)
    net = RevResNet().to(dev).eval()
    net.load_state_dict(synthetic_state_dict())
    w = net._ensure_packed(dev)
    sa, sb = torch.cuda.Stream(device=dev, priority=args.prio), torch.cuda.Stream(device=dev)
    print('stage-3 stream priority', args.prio, torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else '')
    kidx = {(16, 1): 3, (64, 1): 13, (256, 1): 25}
    bufs = {}
    for ch, div in ((16, 1), (64, 2), (256, 4)):
        bufs[ch] = (torch.randn(1, H // div, W // div, ch, device=dev), torch.randn(1, H // div, W // div, ch, device=dev),
                    torch.empty(L.vst_block_tmp_bytes(1, H, W), dtype=torch.uint8, device=dev))

    def blocks(ch, n, stream):
        dst, src, tmp = bufs[ch]
        st = C.c_void_p(stream.cuda_stream)
        for _ in range(n):
            _lib.check(L.vst_block_apply(C.byref(w.blocks[kidx[(ch, 1)]]), ch, 1, 1, _lib.PREC_BF16X3, C.c_void_p(dst.data_ptr()),
                                         C.c_void_p(src.data_ptr()), C.c_void_p(tmp.data_ptr()), 1, H, W, st), "block")

    n = args.reps
    out = []
    for lean in (0, 1):
        _lib.set_option(_lib.OPT_STAGE3_LEAN, lean)
        blocks(256, 3, sa)
        t_a = timed(lambda: blocks(256, n, sa), [sa])
        print(f"stage-3 block, lean {lean}: {t_a / n:7.1f} us alone", flush=True)
        for ch, k in ((16, 2), (64, 2)):                       # k stage-1/2 blocks per stage-3 block: comparable times
            blocks(ch, 3, sb)
            t_b = timed(lambda: blocks(ch, k * n, sb), [sb])

            def both():
                for _ in range(n):
                    blocks(256, 1, sa)
                    blocks(ch, k, sb)
            t_x = timed(both, [sa, sb])
            print(f"    + {k} stage-{1 if ch == 16 else 2} blocks ({t_b / n:6.1f} us alone): together {t_x / n:7.1f} us "
                  f"(sum {(t_a + t_b) / n:6.1f}, max {max(t_a, t_b) / n:6.1f})", flush=True)
            out.append({"lean": lean, "partner_channels": ch, "partner_blocks": k, "stage3_alone_us": round(t_a / n, 1),
                        "partner_alone_us": round(t_b / n, 1), "together_us": round(t_x / n, 1)})
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
