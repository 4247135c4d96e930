"""Same-process A/B of VST_OPT_STAGE3_LEAN (half-CU stage-3 workgroups): bit-identity and frame rates per stream count.

    python tools/lean_ab.py [--size 1024] [--steps 150] [--streams 1,2,3,4] [--pairs 2]

For each stream count the two settings are timed alternately (`--pairs` times each) on the same box; the last line is one
JSON record with every rate, for profiles/.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstnet_amd import _lib                                     # noqa: E402
from vstnet_amd.synth import synthetic_state_dict, synthetic_frames   # noqa: E402
from models.RevResNet import RevResNet                          # noqa: E402
from models.cWCT import cWCT                                    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--streams", default="1,2,3,4")
    ap.add_argument("--pairs", type=int, default=2)
    ap.add_argument("--check-sizes", default="1024x1024,200x280,72x40")
    ap.add_argument("--opt", default="lean", choices=["lean", "pingpong"], help="which option is switched between the two arms")
    ap.add_argument("--steer", action="store_true", help="also time two streams held in anti-phase by events: one stream's "
                    "encode (ending in its stage-3 half) runs against the other's decode (starting with its stage-3 half)")
    args = ap.parse_args()
    OPT = _lib.OPT_STAGE3_LEAN if args.opt == "lean" else _lib.OPT_STAGE3_PINGPONG
    dev = torch.device("cuda", 0)
    net = RevResNet(precision="bf16x3")
    net.load_state_dict(synthetic_state_dict(1234))
    net = net.to(dev).eval()
    cw = cWCT(precision="bf16x3")
    out = {"option": args.opt, "size": args.size, "steps": args.steps, "rates": {}, "bit_identical": {}}
    with torch.no_grad():
        # ---- bit-identity of the two forms (same MFMA order per accumulator) -----------------------------------------
        for spec in args.check_sizes.split(","):
            h, w = (int(v) for v in spec.split("x"))
            xc, xs = synthetic_frames(1, h, w, seed=0).to(dev), synthetic_frames(1, h, w, seed=1).to(dev)
            res = []
            for lean in (0, 1):
                _lib.set_option(OPT, lean)
                zc, zs = net(xc), net(xs)
                zcs = cw.transfer(zc, zs)
                sty = net(zcs, forward=False)
                res.append((torch.as_tensor(zc).clone().float(), sty.clone()))
            torch.cuda.synchronize()
            same = all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
            out["bit_identical"][spec] = bool(same)
            print(f"{args.opt} vs 8-wave at {spec}: bit-identical = {same}", flush=True)
            assert same, spec

        # ---- frame rates ----------------------------------------------------------------------------------------------
        S = args.size
        content = synthetic_frames(1, S, S, seed=0).to(dev)
        style = synthetic_frames(1, S, S, seed=1).to(dev)
        s_stats = cw.style_stats(net(style))

        def frame():
            z = net(content, forward=True)
            return net(cw.transfer_with_stats(z, s_stats), forward=False)

        for ns in (int(v) for v in args.streams.split(",")):
            streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
            for st in streams:                                 # workspaces
                with torch.cuda.stream(st):
                    frame()
            torch.cuda.synchronize()
            for pair in range(args.pairs):
                for lean in (0, 1):
                    _lib.set_option(OPT, lean)
                    for i in range(2 * ns):
                        with torch.cuda.stream(streams[i % ns]):
                            frame()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(args.steps):
                        with torch.cuda.stream(streams[i % ns]):
                            frame()
                    torch.cuda.synchronize()
                    fps = args.steps / (time.perf_counter() - t0)
                    out["rates"].setdefault(f"streams{ns}_lean{lean}", []).append(round(fps, 2))
                    print(f"streams {ns} {args.opt} {lean}: {fps:7.2f} frames/s", flush=True)
        if args.steer:
            sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
            for pair in range(args.pairs):
                for lean in (0, 1):
                    _lib.set_option(OPT, lean)

                    def run(n):
                        # stream a: enc(2k) dec(2k) ...; stream b the same one half-frame later: b's encode of frame k starts
                        # when a's encode of frame k is done, a's encode of frame k+1 when b's encode of frame k is done
                        ev_b = None
                        for _ in range(n // 2):
                            with torch.cuda.stream(sa):
                                if ev_b is not None:
                                    sa.wait_event(ev_b)
                                za = net(content, forward=True)
                                ev_a = torch.cuda.Event()
                                ev_a.record(sa)
                                net(cw.transfer_with_stats(za, s_stats), forward=False)
                            with torch.cuda.stream(sb):
                                sb.wait_event(ev_a)
                                zb = net(content, forward=True)
                                ev_b = torch.cuda.Event()
                                ev_b.record(sb)
                                net(cw.transfer_with_stats(zb, s_stats), forward=False)
                    run(4)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    run(args.steps)
                    torch.cuda.synchronize()
                    fps = args.steps // 2 * 2 / (time.perf_counter() - t0)
                    out["rates"].setdefault(f"steered2_lean{lean}", []).append(round(fps, 2))
                    print(f"two streams in anti-phase, lean {lean}: {fps:7.2f} frames/s", flush=True)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
