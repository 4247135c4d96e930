"""Workload for rocprofv3 --kernel-trace: N stylised 1024x1024 frames on S streams with VST_OPT_STAGE3_LEAN = L.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 tools/trace_frames.py --streams 3 --lean 1
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstnet_amd import _lib                                     # noqa: E402
from vstnet_amd.synth import synthetic_state_dict, synthetic_frames   # noqa: E402
from models.RevResNet import RevResNet                          # noqa: E402
from models.cWCT import cWCT                                    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--frames", type=int, default=24)
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--lean", type=int, default=1)
args = ap.parse_args()
dev = torch.device("cuda", 0)
net = RevResNet(precision="bf16x3")
net.load_state_dict(synthetic_state_dict(1234))
net = net.to(dev).eval()
cw = cWCT(precision="bf16x3")
_lib.set_option(_lib.OPT_STAGE3_LEAN, args.lean)
with torch.no_grad():
    S = args.size
    content = synthetic_frames(1, S, S, seed=0).to(dev)
    style = synthetic_frames(1, S, S, seed=1).to(dev)
    s_stats = cw.style_stats(net(style))
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)]
    for i in range(args.frames):
        with torch.cuda.stream(streams[i % args.streams]):
            z = net(content, forward=True)
            net(cw.transfer_with_stats(z, s_stats), forward=False)
    torch.cuda.synchronize()
print("done")
