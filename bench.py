#!/usr/bin/env python
"""bench.py — stylised frames/s at 1024x1024 on N MI355X GPUs + roofline of the dominant kernel.

A step = one stylised batch per rank: encode the rank's content frame(s) (RevResNet forward), cWCT
against the cached style statistics, decode (RevResNet inverse) — BASELINE.json config 2 at N=1
("single 1024x1024 frame, photorealistic RevResNet+cWCT, fp32 state").  Frames shard across ranks
with no data-path collective (weak scaling: per-GPU work is fixed); torch.distributed (RCCL) is used
only for the barrier and the max-over-ranks clock.  Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N --steps K --warmup W] [--size 1024] [--mode photo|art]
                    [--frames-per-gpu 1] [--recompute-style] [--no-cpu-baseline]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--height", type=int, default=None, help="frame height if not square (e.g. 1080 with --width 1920)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--masked", type=int, default=0, help="K > 0: per-region cWCT with K-label synthetic masks (config 5)")
    ap.add_argument("--mode", default="photo", choices=["photo", "art"])
    ap.add_argument("--frames-per-gpu", type=int, default=1)
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "fp32"])
    ap.add_argument("--recompute-style", action="store_true", help="re-encode + re-factor the style every frame "
                    "(the reference's video loop, video_transfer.py:195) instead of caching it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inplace-cwct", action="store_true", help="overwrite the content code with the transferred code")
    ap.add_argument("--host-pipeline", type=int, default=0, metavar="FRAMES", help="also time FRAMES uint8 frames that "
                    "start and end in host memory through vstnet_amd.pipeline.FramePipeline (PCIe-inclusive rate; "
                    "reported as an extra field, never as `value`)")
    ap.add_argument("--streams", type=int, default=2, help="independent frames in flight per GPU, one HIP stream each "
                    "(the MFMA-bound and the HBM-bound kernels of different frames overlap); 1 = strictly sequential")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from models.RevResNet import RevResNet
    from models.cWCT import cWCT
    from vstnet_amd import _lib
    from vstnet_amd.synth import synthetic_state_dict, synthetic_frames

    from vstnet_amd.sharding import dist_env, shard_range, timed_steps
    rank, local_rank, world = dist_env()
    n_dev = torch.cuda.device_count()
    use_nccl = world <= n_dev                 # fewer GPUs than ranks (rehearsal on a 1-GPU box): ranks share GPUs, gloo
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, f"launch with torch.distributed.run --nproc-per-node {args.gpus}"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank % n_dev)
        if use_nccl:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    S, fpg = args.size, args.frames_per_gpu
    Hf, Wf = (args.height or S), (args.width or S)
    Hf, Wf = Hf // 4 * 4, Wf // 4 * 4                     # img_resize floors to a multiple of down_scale = 4
    hd, sp = (16, 2) if args.mode == "photo" else (64, 1)
    sd = synthetic_state_dict(1234, hd, sp)
    net = RevResNet(hidden_dim=hd, sp_steps=sp, precision=args.precision)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    cw = cWCT()

    # frame f of the job has seed (0, f); rank r owns a contiguous shard of the world*fpg frames
    first, last = shard_range(world * fpg, rank, world)
    import numpy as np
    frames = []
    for f in range(first, last):
        rng = np.random.Generator(np.random.PCG64([0, f]))
        frames.append(rng.random((3, Hf, Wf), dtype=np.float32))
    content = torch.from_numpy(np.stack(frames)).to(dev)
    style = synthetic_frames(1, Hf, Wf, seed=1).to(dev)
    cmask = smask = None
    if args.masked:
        from vstnet_amd.synth import synthetic_mask
        assert args.mode == "photo", "this fork's masked cWCT needs masks at code resolution (photorealistic codes)"
        cmask = np.stack([synthetic_mask(Hf, Wf, args.masked, seed=3)] * fpg)
        smask = np.stack([synthetic_mask(Hf, Wf, args.masked, seed=4, speck=False)] * fpg)

    with torch.no_grad():
        z_s = net(style)
        s_stats = cw.style_stats(z_s)
        plan = None
        if args.masked and not args.recompute_style:      # the masks and the style do not change between frames
            plan = cw.bind_style(cw.plan_masks(cmask, smask, (fpg,) + tuple(z_s.shape[1:]), (fpg,) + tuple(z_s.shape[1:]), dev),
                                 z_s.expand(fpg, -1, -1, -1))

        def stylize_batch():
            z_c = net(content, forward=True)
            if args.masked and plan is not None:
                z_cs = cw.transfer_with_plan(z_c, None, plan)
            elif args.masked:
                zs = net(style, forward=True)
                z_cs = cw.transfer(z_c, zs.expand(fpg, -1, -1, -1), cmask, smask)
            elif args.recompute_style:
                zs = net(style, forward=True)
                z_cs = cw.transfer(z_c, zs.expand(fpg, -1, -1, -1))
            else:
                z_cs = cw.transfer_with_stats(z_c, s_stats, inplace=args.inplace_cwct)
            return net(z_cs, forward=False)

        # every step is one independent batch; consecutive steps alternate over `--streams` HIP streams so that
        # up to that many frames are in flight (all inputs / style statistics are ready before the timed region)
        streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]
        torch.cuda.synchronize()
        counter = [0]

        def step():
            st = streams[counter[0] % len(streams)]
            counter[0] += 1
            with torch.cuda.stream(st):
                return stylize_batch()

        elapsed = timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, world, device=dev if use_nccl else "cpu")
        assert torch.isfinite(step()).all()

        # ---- live roofline of the dominant kernel: HIP events around each of its launches ------------
        L = _lib.lib()
        cin, cout = 256, 64                         # stage-3 / channel_reduction conv.1 (largest share of MFMA work)
        _lib.check(L.vst_profile_begin(_lib.kernel_id(cin, cout, 1), 4096), "vst_profile_begin")
        for _ in range(max(1, min(args.steps, 10))):     # one frame at a time on one stream: with frames in flight on two
            with torch.cuda.stream(streams[0]):          # streams the events would also time the other frame's kernels
                stylize_batch()
            streams[0].synchronize()
        tot_ms, n_launch = C.c_double(0), C.c_int(0)
        _lib.check(L.vst_profile_end(C.byref(tot_ms), C.byref(n_launch)), "vst_profile_end")
        avg_ms = tot_ms.value / max(1, n_launch.value)
        px = fpg * (Hf // 4) * (Wf // 4)
        alg_flops = 2.0 * 9 * cin * cout * px        # fp32-equivalent conv flops (the split executes 3x as bf16 MFMA)
        achieved_tf = alg_flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0

    # HBM bytes per launch of that kernel from the committed PMC summary (separate rocprofv3 --pmc passes of this same
    # command; FETCH_SIZE corrected x2 for gfx950) — only valid for the workload it was taken on
    traffic = None
    pmc_path = os.path.join(REPO, "profiles", "r01_pmc_hbm_traffic.json")
    if os.path.exists(pmc_path) and (Hf, Wf) == (1024, 1024) and fpg == 1 and args.mode == "photo":
        k = json.load(open(pmc_path))["kernels"].get(f"void conv_pipe_kernel<{cin}, {cout}, true, false>(ConvArgs)")
        traffic = k["hbm_bytes_per_launch"] if k else None

    frames_total = world * fpg * args.steps
    ms_per_step = elapsed / args.steps * 1e3
    value = frames_total / elapsed
    passes = 3 if args.recompute_style else 2
    frame_bytes = (passes * 6540 + 384) * Hf * Wf + (128 * Hf * Wf if args.recompute_style else 0)
    frame_gbs = frame_bytes * fpg * args.steps / elapsed / 1e9     # per GPU

    rec = {
        "metric": "stylized frames/sec at 1024x1024 (1/2/4/8 GPU) + % HBM roofline",
        "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (bf16x3 split MFMA, f32 accumulate/state)" if args.precision == "bf16x3" else "f32",
        "data": "synthetic",
        "config": {"workload": f"{'photorealistic' if args.mode == 'photo' else 'artistic'} {Wf}x{Hf} frame: RevResNet "
                   f"forward + cWCT ({str(args.masked) + '-label masked, ' if args.masked else ''}{'style re-encoded per frame' if args.recompute_style else 'style statistics cached'})"
                   " + RevResNet inverse", "frames_per_gpu": fpg, "sharding": f"{world} ranks x {fpg} frame(s), no collective",
                   "weights": "synthetic seed 1234", "frames_in_flight": max(1, args.streams)},
        "roofline": {"kernel": f"conv_pipe_kernel<{cin},{cout}> (stage-3 / channel_reduction conv.1)", "bound": "mfma",
                     "achieved": round(achieved_tf, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved_tf / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic,
                     "avg_launch_ms": round(avg_ms, 5), "launches_timed": n_launch.value,
                     "note": "achieved counts algorithmic fp32 conv flops; the bf16x3 split issues 3x that on the MFMA pipe"},
        "frame_hbm_roofline": {"bound": "hbm", "algorithmic_bytes_per_frame": frame_bytes,
                               "achieved": round(frame_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(frame_gbs / HBM_PEAK_GBS, 4), "per": "GPU"},
    }

    if args.host_pipeline > 0 and not args.masked and fpg == 1:
        # PCIe-inclusive: uint8 frames in pageable host memory -> pinned ring -> H2D -> encode/cWCT/decode -> D2H -> host
        from vstnet_amd.pipeline import FramePipeline
        with torch.no_grad():
            host = [(content[0].permute(1, 2, 0) * 255).byte().contiguous().cpu().numpy()] * 4
            pipe = FramePipeline(net, lambda z, i: cw.transfer_with_stats(z, s_stats), Hf, Wf, device=dev, depth=4,
                                 compute_streams=max(1, args.streams))
            pipe.run((host[i % 4] for i in range(8)), lambda i, a: None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe.run((host[i % 4] for i in range(args.host_pipeline)), lambda i, a: None)
            dt = time.perf_counter() - t0
        rec["host_pipeline"] = {"value": round(args.host_pipeline / dt, 2), "unit": "frames/s", "per": "GPU",
                                "frames": args.host_pipeline, "note": "uint8 HWC frames from and to host memory, pinned "
                                "ring buffers, H2D/compute/D2H overlapped; PCIe-inclusive, not `value`"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        rec["cpu_baseline"] = cpu_baseline(sd, sp, Hf, Wf)
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


def usable_cores():
    """CPU share of this process: affinity mask, capped by the cgroup quota (the GPU box exposes every host
    core in os.cpu_count() but schedules the job on a 16-core share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("VST_CPU_THREADS", "16"))))


def cpu_baseline(sd, sp, H, W):
    """The oracle (CPU restatement of the reference's torch-op sequence) timed on this host's cores on ONE
    frame of the same workload (style code precomputed, like the GPU leg)."""
    import torch
    from oracle import cpu_ref
    from vstnet_amd.synth import synthetic_frames
    cores = usable_cores()
    torch.set_num_threads(cores)
    xc, xs = synthetic_frames(1, H, W, seed=0), synthetic_frames(1, H, W, seed=1)
    with torch.no_grad():
        small = synthetic_frames(1, 64, 64, seed=2)
        cpu_ref.revnet_inverse(cpu_ref.revnet_forward(small, sd, sp), sd, sp)       # warm the thread pool
        zs = cpu_ref.revnet_forward(xs, sd, sp)
        t0 = time.perf_counter()
        zc = cpu_ref.revnet_forward(xc, sd, sp)
        zcs = cpu_ref.transfer(zc, zs)
        cpu_ref.revnet_inverse(zcs, sd, sp)
        dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 frame {W}x{H} (forward + cWCT + inverse, style code precomputed), oracle/cpu_ref.py on torch CPU ops",
            "seconds": round(dt, 2)}


if __name__ == "__main__":
    main()
