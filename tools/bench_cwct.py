"""Time the cWCT statistics / apply entry points alone (unmasked vs the single-pass label-slot forms)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstnet_amd import _lib                      # noqa: E402
from vstnet_amd.cwct import cWCT                 # noqa: E402
from vstnet_amd.synth import synthetic_mask      # noqa: E402


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--channels", type=int, default=32)
    ap.add_argument("--labels", type=int, default=5)
    args = ap.parse_args()
    H, W, N = args.height, args.width, args.channels
    dev = torch.device("cuda", 0)
    cw = cWCT()
    z = torch.randn(1, N, H, W, device=dev)
    zs = torch.randn(1, N, H, W, device=dev) * 0.5 + 0.1
    x2 = z.reshape(N, -1)
    print(f"unmasked stats {timeit(lambda: cw.stats(x2)):8.1f} us   apply {timeit(lambda: cw.apply(x2, cw.factor(cw.stats(x2), [cw.stats(zs.reshape(N, -1))], [1.0], 0.0, N))):8.1f} us (incl. stats x2 + factor)")
    for kind in ("bands", "noise"):
        cm = synthetic_mask(H, W, args.labels, seed=3, kind=kind)[None]
        sm = synthetic_mask(H, W, args.labels, seed=4, speck=False, kind=kind)[None]
        plan = cw.learn_slots(cw.plan_masks(cm, sm, z.shape, zs.shape, dev))
        t_stats = timeit(lambda: cw._stats_labels(x2, plan.cm[0], plan.tables[0], plan.max_slots))
        bound = cw.bind_style(plan, zs)
        t_all = timeit(lambda: cw.transfer_with_plan(z, None, bound))
        print(f"{kind:6s} labels={args.labels}: stats_labels {t_stats:8.1f} us   transfer_with_plan {t_all:8.1f} us")


if __name__ == "__main__":
    main()
