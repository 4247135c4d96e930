#!/usr/bin/env python
"""bench.py — stylised frames/s at 1024x1024 on N MI355X GPUs + roofline of the dominant kernel.

A step = one stylised batch per rank: encode the rank's content frame(s) (RevResNet forward), cWCT
against the cached style statistics, decode (RevResNet inverse) — BASELINE.json config 2 at N=1
("single 1024x1024 frame, photorealistic RevResNet+cWCT, fp32 state").  Frames shard across ranks
with no data-path collective (weak scaling: per-GPU work is fixed); torch.distributed (RCCL) is used
only for the barrier and the max-over-ranks clock.  Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N --steps K --warmup W] [--size 1024] [--mode photo|art]
                    [--frames-per-gpu 1] [--recompute-style] [--no-cpu-baseline]

`python bench.py --gpus N` (N > 1) without a launcher starts its own N ranks (one child process per rank,
before this process touches a GPU); under `python -m torch.distributed.run --nproc-per-node N` it uses the
ranks it is given.  Rank 0 prints the one JSON line.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_COPY_GBS = 6290.0          # the same guide's measured float4 copy rate: SURVEY 8(d)'s "second denominator"
MFMA_16BIT_PEAK_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA peak
PMC_FILE = "r04_pmc_hbm_traffic.json"   # committed rocprofv3 --pmc summary that `roofline.traffic` is read from (never measured live)

# conv kernel classes by profile id (cin, cout, stride) -> (stage, output-pixel divisor w.r.t. the full-resolution frame,
# multiply-accumulates per output pixel and launch); the (4,16) and (16,64) ids are the fused conv.4 + conv.7 launches
# algorithmic HBM bytes per OUTPUT pixel of a launch at 4 B per value (fp32, or an fp16 hi + lo pair): input read (x stride^2
# for the stride-2 convs) + output written (+ the old state read where the conv ends a coupling block); DESIGN.md kernel table
CONV_BYTES = {
    (16, 4, 1): 64 + 16, (4, 16, 1): 16 + 64 + 64, (16, 16, 2): 4 * 64 + 64, (64, 16, 1): 256 + 64, (16, 64, 1): 64 + 256 + 256,
    (64, 64, 2): 4 * 256 + 256, (256, 64, 1): 1024 + 256, (64, 64, 1): 256 + 256, (64, 256, 1): 256 + 1024 + 1024,
}
CONV_CLASSES = {
    (16, 4, 1): ("stage1", 1, 16 * 4), (4, 16, 1): ("stage1", 1, 4 * 4 + 4 * 16),
    (16, 16, 2): ("stage2", 4, 16 * 16), (64, 16, 1): ("stage2", 4, 64 * 16), (16, 64, 1): ("stage2", 4, 16 * 16 + 16 * 64),
    (64, 64, 2): ("stage3", 16, 64 * 64), (256, 64, 1): ("stage3", 16, 256 * 64), (64, 64, 1): ("stage3", 16, 64 * 64),
    (64, 256, 1): ("stage3", 16, 64 * 256),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # ~0.85 s timed: sustained clocks, not a burst
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--height", type=int, default=None, help="frame height if not square (e.g. 1080 with --width 1920)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--masked", type=int, default=0, help="K > 0: per-region cWCT with K-label synthetic masks (config 5)")
    ap.add_argument("--mask-kind", default="bands", choices=["bands", "noise"], help="synthetic label maps: vertical bands, or "
                    "per-pixel random labels (the worst case for any per-label tile skipping)")
    ap.add_argument("--mode", default="photo", choices=["photo", "art"])
    ap.add_argument("--frames-per-gpu", type=int, default=1)
    ap.add_argument("--precision", default=None, choices=["f16x2", "f16x2h", "bf16x3", "fp32", "auto"],
                    help="conv arithmetic of the timed region (`value`); default: the library's default = bf16x3, the fp32-class "
                         "mode.  The fp16 modes are opt-in and reported per mode under `modes` with their own parity figures")
    ap.add_argument("--recompute-style", action="store_true", help="re-encode + re-factor the style every frame "
                    "(the reference's video loop, video_transfer.py:195) instead of caching it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-stage table and the recompute-style rate")
    ap.add_argument("--inplace-cwct", action="store_true", help="overwrite the content code with the transferred code")
    ap.add_argument("--host-pipeline", type=int, default=0, metavar="FRAMES", help="also time FRAMES uint8 frames that "
                    "start and end in host memory through vstnet_amd.pipeline.FramePipeline (PCIe-inclusive rate; "
                    "reported as an extra field, never as `value`)")
    ap.add_argument("--streams", type=int, default=3, help="independent frames in flight per GPU, one HIP stream each "
                    "(the MFMA-bound and the HBM-bound kernels of different frames overlap); 1 = strictly sequential")
    ap.add_argument("--stage3-lean", type=int, default=None, choices=[0, 1], help="VST_OPT_STAGE3_LEAN (vstnet.h): the bf16x3 256-channel "
                    "convs as half-CU workgroups (measured: slower, profiles/r04_lean_overlap.json; default: the library's, 0)")
    ap.add_argument("--dry-run-ms", type=float, default=0.0, help="host-logic rehearsal (tests): no GPU work, a step sleeps "
                    "this many milliseconds; exercises launch, rendezvous, timing and the JSON line only")
    return ap.parse_args(argv)


def launch_ranks(n, argv):
    """Start one child per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* and a CPU thread cap of cores // n in its
    environment) and wait for all of them.  The parent never initialises a GPU; a failed rank takes the others down and its
    exit code is returned."""
    from vstnet_amd.sharding import launch_children, rank_environment
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    return launch_children([cmd] * n, [rank_environment(r, n, port) for r in range(n)])


def main():
    args = parse_args()
    # decide on self-launch before anything touches the GPU (the children do; this process never does)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from vstnet_amd.sharding import dist_env, shard_range, timed_steps
    rank, local_rank, world = dist_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}")
    if args.dry_run_ms > 0:
        return dry_run(args, rank, world)

    from models.RevResNet import RevResNet
    from models.cWCT import cWCT
    from vstnet_amd import _lib
    from vstnet_amd.synth import synthetic_state_dict, synthetic_frames
    args.precision = args.precision or _lib.default_precision()
    if args.stage3_lean is not None:
        _lib.set_option(_lib.OPT_STAGE3_LEAN, args.stage3_lean)
    n_dev = torch.cuda.device_count()          # counting devices does not initialise HIP
    use_nccl = world <= n_dev                 # fewer GPUs than ranks (rehearsal on a 1-GPU box): ranks share GPUs, gloo
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank % n_dev)
        from vstnet_amd.sharding import rank_threads
        torch.set_num_threads(rank_threads(world))        # the ranks of a node share its cores (launch_ranks caps OMP too)
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
    cw = cWCT(precision=args.precision)

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
        cmask = np.stack([synthetic_mask(Hf, Wf, args.masked, seed=3, kind=args.mask_kind)] * fpg)
        smask = np.stack([synthetic_mask(Hf, Wf, args.masked, seed=4, speck=False, kind=args.mask_kind)] * fpg)

    with torch.no_grad():
        z_s = net(style)
        s_stats = cw.style_stats(z_s)
        plan = None
        if args.masked and not args.recompute_style:      # the masks and the style do not change between frames
            plan = cw.plan_masks(cmask, smask, (fpg,) + tuple(z_s.shape[1:]), (fpg,) + tuple(z_s.shape[1:]), dev)
            plan = cw.bind_style(cw.learn_slots(plan), z_s.expand(fpg, -1, -1, -1))

        def stylize_batch(recompute=args.recompute_style, keep=None):
            z_c = net(content, forward=True)
            if args.masked and plan is not None and not recompute:
                z_cs = cw.transfer_with_plan(z_c, None, plan)
            elif args.masked:
                zs = net(style, forward=True)
                z_cs = cw.transfer(z_c, zs.expand(fpg, -1, -1, -1), cmask, smask)
            elif recompute:
                zs = net(style, forward=True)
                z_cs = cw.transfer(z_c, zs.expand(fpg, -1, -1, -1))
            else:
                z_cs = cw.transfer_with_stats(z_c, s_stats, inplace=args.inplace_cwct and keep is None)
            out = net(z_cs, forward=False)
            if keep is not None:
                keep.update(z_c=z_c, z_cs=z_cs, stylized=out)
            return out

        # every step is one independent batch; consecutive steps alternate over `--streams` HIP streams so that
        # up to that many frames are in flight (all inputs / style statistics are ready before the timed region)
        streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]
        torch.cuda.synchronize()
        counter = [0]

        def step(**kw):
            st = streams[counter[0] % len(streams)]
            counter[0] += 1
            with torch.cuda.stream(st):
                return stylize_batch(**kw)

        # setup, not warm-up: one untimed step per stream, so that every stream's pass workspace (288 B per pixel: 4.8 GB at
        # 4096 x 4096) exists before the W warm-up steps — with W < streams an allocation would otherwise land in the timed region
        for _ in streams:
            step()
        torch.cuda.synchronize()
        elapsed, per_rank_s = timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, world,
                                          device=dev if use_nccl else "cpu", return_all=True)
        assert torch.isfinite(step()).all()
        torch.cuda.synchronize()
        args.effective_precision = net.resolved_precision

        # ---- extras, all outside the timed region ----------------------------------------------------------------------
        extras = {}
        table = {}
        n_prof = max(1, min(args.steps, 5))
        if rank == 0:
            # HIP events around every launch, one frame at a time on one stream (with frames in flight on two streams the
            # events would also time the other frame's kernels)
            def prof_frames():
                for _ in range(n_prof):
                    with torch.cuda.stream(streams[0]):
                        stylize_batch()
                    streams[0].synchronize()
            table = _lib.profile_table(prof_frames)
        if rank == 0 and not args.no_extras and not args.masked:
            t3 = timed_steps(lambda: step(recompute=True), max(2, min(args.steps, 10)), 2, torch.cuda.synchronize, 1)
            n3 = max(2, min(args.steps, 10))
            extras["recompute_style"] = {
                "value": round(fpg * n3 / t3, 3), "unit": "frames/s", "per": "GPU", "passes": 3,
                "note": "style re-encoded and re-factored for every frame (the reference's video loop, video_transfer.py:195): "
                        "3 RevResNet passes + style statistics per frame"}
        # ---- per arithmetic mode: rate, sequential rate, frame roofline and (below, against the CPU leg's oracle frame) parity --
        mode_rates, mode_out = {}, {}
        if rank == 0 and world == 1 and not args.no_extras and not args.masked and not args.recompute_style:
            n_alt = max(2, min(args.steps, 60))
            for p_alt in ("bf16x3", "f16x2", "f16x2h", "auto"):
                if p_alt == args.precision:
                    net_a, s_alt = net, s_stats
                else:
                    net_a = RevResNet(hidden_dim=hd, sp_steps=sp, precision=p_alt)
                    net_a.load_state_dict(sd)
                    net_a = net_a.to(dev).eval()
                    s_alt = cw.style_stats(net_a(style))

                def frame_alt(keep=None):
                    z_c = net_a(content, forward=True)
                    z_cs = cw.transfer_with_stats(z_c, s_alt)
                    out = net_a(z_cs, forward=False)
                    if keep is not None:
                        keep.update(z_c=z_c, z_cs=z_cs, stylized=out)
                    return out

                def step_alt():
                    st = streams[counter[0] % len(streams)]
                    counter[0] += 1
                    with torch.cuda.stream(st):
                        return frame_alt()

                def step_seq():
                    with torch.cuda.stream(streams[0]):
                        return frame_alt()
                t_alt = timed_steps(step_alt, n_alt, 3, torch.cuda.synchronize, 1)
                t_seq = timed_steps(step_seq, n_alt, 2, torch.cuda.synchronize, 1)
                mode_rates[p_alt] = (fpg * n_alt / t_alt, fpg * n_alt / t_seq)
                if p_alt == "auto":
                    extras["auto_calibration"] = net_a.calibration
                if not args.no_cpu_baseline:
                    keep = {}
                    frame_alt(keep)
                    torch.cuda.synchronize()
                    mode_out[p_alt] = {k: v[:1].float().cpu() for k, v in keep.items()}
                if net_a is not net:
                    del net_a
        gpu_out = {}
        if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.masked:
            if args.precision in mode_out:
                gpu_out = mode_out[args.precision]
            else:
                stylize_batch(recompute=False, keep=gpu_out)
                torch.cuda.synchronize()
                gpu_out = {k: v[:1].float().cpu() for k, v in gpu_out.items()}
        range_flags = _lib.range_flags(reset=False) if rank == 0 else 0

    frames_total = world * fpg * args.steps
    ms_per_step = elapsed / args.steps * 1e3
    value = frames_total / elapsed
    passes = 3 if args.recompute_style else 2
    frame_bytes = (passes * 6540 + 384) * Hf * Wf + (128 * Hf * Wf if args.recompute_style else 0)
    frame_gbs = frame_bytes * fpg * args.steps / elapsed / 1e9     # per GPU
    per_rank_fps = [round(fpg * args.steps / t, 3) for t in per_rank_s]

    prec_note = {"f16x2": "f32 state / f32 accumulate; fp16 2-term split MFMA: w_hi (x_hi + x_lo), weights rounded to fp16 (opt-in mode)",
                 "f16x2h": "f32 state (fp16 hi+lo planes in the 256-channel blocks) / f32 accumulate; fp16 MFMA, weights rounded to "
                           "fp16; conv inputs that cross HBM (h1, h2, the state as the 256-channel blocks' first conv reads it) are "
                           "fp16 tensors: 1 MFMA per product; operands split in-kernel from the f32 state: fp16 hi+lo, 2 MFMAs (opt-in mode)",
                 "bf16x3": "f32 state / f32 accumulate; bf16 3-term split MFMA (a_hi w_hi + a_lo w_hi + a_hi w_lo): fp32-class, "
                           "3e-6 of the reference; the library default", "fp32": "f32",
                 "auto": "self-calibrated: one of f16x2h / f16x2 / bf16x3, see auto_calibration"}[args.precision]
    rec = {
        "metric": "stylized frames/sec at 1024x1024 (1/2/4/8 GPU) + % HBM roofline",
        "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": (net.resolved_precision or args.precision), "dtype_note": prec_note, "data": "synthetic",
        "config": {"workload": f"{'photorealistic' if args.mode == 'photo' else 'artistic'} {Wf}x{Hf} frame: RevResNet "
                   f"forward + cWCT ({str(args.masked) + '-label ' + args.mask_kind + ' masks, ' if args.masked else ''}{'style re-encoded per frame' if args.recompute_style else 'style statistics cached'})"
                   " + RevResNet inverse", "frames_per_gpu": fpg, "sharding": f"{world} ranks x {fpg} frame(s), no collective",
                   "weights": "synthetic seed 1234", "frames_in_flight": max(1, args.streams), "precision": args.precision,
                   "stage3_lean": _lib.get_option(_lib.OPT_STAGE3_LEAN),
                   "resolved_precision": net.resolved_precision},
        "per_rank": {"frames_per_s": per_rank_fps, "min": min(per_rank_fps), "max": max(per_rank_fps)},
        "frame_hbm_roofline": {"bound": "hbm", "algorithmic_bytes_per_frame": frame_bytes,
                               "achieved": round(frame_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(frame_gbs / HBM_PEAK_GBS, 4), "frac_of_measured_copy_rate": round(frame_gbs / HBM_COPY_GBS, 4),
                               "measured_copy_rate": HBM_COPY_GBS, "per": "GPU"},
    }
    if rank == 0 and table:
        rec["roofline"], rec["stages"], rec["kernel_classes"], rec["frame_level"] = roofline_from_table(
            table, n_prof, fpg, Hf, Wf, args, _lib, ms_per_step=ms_per_step / fpg, frames_in_flight=max(1, args.streams))
    rec.update(extras)
    if rank == 0:
        rec["fp16_range_flags"] = {"value": int(range_flags), "note": "device-side flags since process start (vstnet.h VST_RANGE_*: "
                                   "1 = an activation saturated at +-65504 in an fp16 mode, 2 = a weight beyond fp16); 0 = none"}
        rec["timed_region_note"] = ("every step re-encodes the same content tensor: the working set of a frame (288 B/px of "
                                    "workspace + code = 0.4 GB at 1024x1024, x frames in flight) is far beyond the 256 MiB "
                                    "Infinity Cache, so HBM traffic per step equals a fresh frame's")
    if mode_rates:
        per_frame_bytes = (2 * 6540 + 384) * Hf * Wf
        rec["modes"] = {m: {"frames_per_s": round(v[0], 3), "sequential_frames_per_s": round(v[1], 3),
                            "frame_hbm_roofline_frac": round(per_frame_bytes * v[0] / 1e9 / HBM_PEAK_GBS, 4)}
                        for m, v in mode_rates.items()}
        rec["modes"]["note"] = ("same workload, same streams, per arithmetic mode; sequential = one HIP stream, one frame at a time; "
                                "parity_* (added by the CPU leg) = this run's frame 0 vs the oracle frame; `value` is the mode "
                                f"named in config.precision ({args.precision})")
        rec["sequential_frames_per_s"] = rec["modes"].get(args.precision, {}).get("sequential_frames_per_s")
        if "auto" in rec["modes"] and "auto_calibration" in rec:
            rec["modes"]["auto"]["resolved"] = rec["auto_calibration"]["chosen"]
            rec["modes"]["auto"]["what"] = ("precision='auto' (opt-in): the fastest mode whose probe stylisation on this checkpoint "
                                            "stays within auto_calibration.tolerance of bf16x3 on the device")

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
        rec["cpu_baseline"] = cpu_baseline(sd, sp, Hf, Wf, gpu_out, mode_out=mode_out, modes_rec=rec.get("modes"))
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


def conv_terms(f16, cin, cout, stride):
    """MFMA products issued per algorithmic product: 2 (fp16 2-term) in the f16x2 modes - 1 in the 256-channel blocks' last conv
    under f16x2h, whose inputs are fp16 planes -, 3 (bf16 3-term) in the bf16x3 mode."""
    if f16 == "h" and cin >= 64 and cout >= 64 and stride == 1:
        return 1
    return 2 if f16 else 3


def kernel_label(cin, cout, stride, terms):
    if cin >= 64 and cout >= 64 and stride == 1:
        return (f"conv_sp_kernel<{cin},{cout}> (fp16 {terms}-term, LDS-DMA)" if terms <= 2
                else f"conv_pipe_kernel<{cin},{cout}> (bf16 3-term)")
    fused = (cin, cout) in ((4, 16), (16, 64))
    return (f"{'conv_pair_kernel' if fused else 'conv_mfma_kernel'}<{cin},{cout}{',s2' if stride == 2 else ''}> "
            f"({'fp16 2-term' if terms == 2 else 'bf16 3-term'}{', conv.4 + conv.7 fused' if fused else ''})")


# profile-table ids of the non-conv launches -> the kernels of the PMC summary that make up such a launch
MISC_PMC = {1: ("pack_input_kernel", "block0_const_kernel"), 2: ("unpack_output_kernel",), 3: ("spread_gather_kernel",),
            4: ("spread_gather_kernel",), 5: ("cwct_stats_pm_kernel", "cwct_stats_finish_kernel"), 6: ("cwct_factor_kernel",),
            7: ("cwct_apply_pm_kernel",), 8: ("presplit_kernel",)}


def pmc_lookup(pmc, cin, cout, stride):
    """HBM bytes per launch of a conv class from the committed PMC summary (None if the class is not in it)"""
    if cin >= 64 and cout >= 64 and stride == 1:
        pats = [f"conv_pipe_kernel<{cin}, {cout},", f"conv_sp_kernel<{cin}, {cout},"]
    elif (cin, cout) in ((4, 16), (16, 64)) and stride == 1:
        pats = [f"conv_pair_kernel<{cin}, {cout},"]
    else:
        pats = [f"conv_mfma_kernel<{cin}, {cout}, {stride},"]
    hit = [v for n, v in pmc["kernels"].items() if any(t in n for t in pats)]
    return hit[0]["hbm_bytes_per_launch"] if hit else None


def roofline_from_table(table, n_frames, fpg, H, W, args, _lib, ms_per_step=None, frames_in_flight=1):
    """From the HIP-event table of `n_frames` frames run one at a time (ms are per frame below): the per-launch roofline of the
    conv class with the largest total time (BOTH roofs; the binding one is decided on the ISSUED matrix work: terms x flops),
    the five classes with the largest totals, a per-stage summary, and the frame-level picture (HBM bytes per frame by the
    committed PMC summary, chip-average TB/s and issued PFLOP/s at the timed region's ms per step)."""
    px = fpg * H * W                                   # full-resolution pixels per frame batch
    prec = getattr(args, "effective_precision", None) or args.precision
    f16 = "h" if prec == "f16x2h" else prec == "f16x2"      # truthy for both fp16 modes
    per = {}                                           # (cin, cout, stride) -> (ms per frame, launches per frame)
    for kid, (ms, cnt) in table.items():
        if kid >= 65536:
            per[(kid >> 16, (kid >> 4) & 0xFFF, kid & 15)] = (ms / n_frames, cnt / n_frames)
    per = {k: v for k, v in per.items() if k in CONV_CLASSES}
    # `traffic` is NOT measured in this run: rocprofv3 PMC counters need their own passes (separate --pmc runs); it is read from
    # the committed summary of such a run of this same command, and `traffic_source` says which file, for which precision, from
    # which commit — a kernel edited since then makes it stale, which the commit hash shows.
    pmc, traffic_source = None, None
    pmc_path = os.path.join(REPO, "profiles", PMC_FILE)
    if os.path.exists(pmc_path) and (H, W) == (1024, 1024) and fpg == 1 and args.mode == "photo":
        cand = json.load(open(pmc_path))
        if cand.get("precision", "f16x2h") == prec:
            pmc = cand
            traffic_source = {"file": "profiles/" + PMC_FILE, "precision": pmc.get("precision", "f16x2h"),
                              "git_commit": pmc.get("git_commit"), "collected": "rocprofv3 --pmc, separate passes (not this run)"}

    def klass(key):
        """both roofs of one conv class"""
        cin, cout, stride = key
        ms, cnt = per[key]
        _, div, macs = CONV_CLASSES[key]
        terms = conv_terms(f16, cin, cout, stride)
        flops = 2.0 * 9 * macs * px / div              # fp32-equivalent conv flops of one launch
        avg_ms = ms / cnt
        per_px = CONV_BYTES[key]
        if f16 == "h":                                     # fp16 tensors where f16x2 moves hi + lo pairs (or fp32)
            per_px -= {(256, 64, 1): 512 + 128, (64, 64, 1): 128 + 128, (64, 256, 1): 128,  # state hi plane, h1, h2 of the 256-channel blocks
                       (16, 4, 1): 8, (4, 16, 1): 8, (64, 16, 1): 32, (16, 64, 1): 32, (16, 16, 2): 32}.get(key, 0)
        nbytes = per_px * px / div
        traffic = pmc_lookup(pmc, cin, cout, stride) if pmc else None
        t = avg_ms * 1e-3
        t_hbm, t_mfma_issued = nbytes / (HBM_PEAK_GBS * 1e9), flops * terms / (MFMA_16BIT_PEAK_TFLOPS * 1e12)
        rec = {"kernel": kernel_label(cin, cout, stride, terms), "launches_per_frame": round(cnt, 1), "avg_launch_us": round(avg_ms * 1e3, 2),
               "ms_per_frame": round(ms, 4),
               "hbm": {"algorithmic_bytes": int(nbytes), "algorithmic_GBps": round(nbytes / t / 1e9, 1),
                       "frac": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4), "pmc_bytes": traffic,
                       "pmc_GBps": round(traffic / t / 1e9, 1) if traffic else None,
                       "pmc_over_algorithmic": round(traffic / nbytes, 3) if traffic else None},
               "mfma": {"algorithmic_flops": int(flops), "terms": terms, "algorithmic_TFLOPps": round(flops / t / 1e12, 2),
                        "issued_TFLOPps": round(flops * terms / t / 1e12, 2),
                        "frac": round(flops / t / 1e12 / MFMA_16BIT_PEAK_TFLOPS, 4),
                        "issued_frac": round(flops * terms / t / 1e12 / MFMA_16BIT_PEAK_TFLOPS, 4)},
               "min_us_at_peak": {"hbm": round(t_hbm * 1e6, 2), "mfma_issued": round(t_mfma_issued * 1e6, 2)},
               "bound": "mfma" if t_mfma_issued > t_hbm else "hbm"}
        return rec

    classes = {k: klass(k) for k in per}
    top = sorted(classes, key=lambda k: -per[k][0])
    dom = classes[top[0]]
    roof = {"kernel": dom["kernel"] + " — the conv class with the largest total time per frame",
            "bound": dom["bound"], "traffic": dom["hbm"]["pmc_bytes"], "traffic_source": traffic_source,
            "avg_launch_ms": round(dom["avg_launch_us"] * 1e-3, 5), "launches_per_frame": dom["launches_per_frame"],
            "ms_per_frame": dom["ms_per_frame"],
            "both_roofs": {"hbm": dom["hbm"], "mfma": dom["mfma"], "min_us_at_peak": dom["min_us_at_peak"]},
            "note": "the binding roof is the one with the larger minimum time for the launch: ALGORITHMIC bytes at the HBM peak against the "
                    f"ISSUED matrix work ({dom['mfma']['terms']} MFMA terms per product x 2*9*cin*cout flops per output pixel) at the dense "
                    "16-bit MFMA peak; HIP events on the launch stream, one frame at a time"}
    if dom["bound"] == "hbm":
        roof.update({"achieved": dom["hbm"]["algorithmic_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["hbm"]["frac"],
                     "mfma_issued_frac": dom["mfma"]["issued_frac"]})
    else:
        roof.update({"achieved": dom["mfma"]["issued_TFLOPps"], "peak": MFMA_16BIT_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": dom["mfma"]["issued_frac"], "achieved_is": "issued MFMA flops (terms x algorithmic) per second",
                     "algorithmic_frac": dom["mfma"]["frac"], "hbm_frac": dom["hbm"]["frac"]})
    # ---- per-stage summary ------------------------------------------------------------------------------------------------
    stage_ms = {"stage1": 0.0, "stage2": 0.0, "stage3": 0.0, "cwct": 0.0, "glue": 0.0}
    stage_issued = {"stage1": 0.0, "stage2": 0.0, "stage3": 0.0}
    stage_pmc = {"stage1": 0.0, "stage2": 0.0, "stage3": 0.0, "cwct": 0.0, "glue": 0.0}
    pmc_complete = pmc is not None
    for (ci, co, st_), (m, c) in per.items():
        stg, dv, mc = CONV_CLASSES[(ci, co, st_)]
        stage_ms[stg] += m
        t = conv_terms(f16, ci, co, st_)
        stage_issued[stg] += 2.0 * 9 * mc * px / dv * c * t
        tb = classes[(ci, co, st_)]["hbm"]["pmc_bytes"]
        if tb is None:
            pmc_complete = False
        else:
            stage_pmc[stg] += tb * c
    for kid, (m, c) in table.items():
        if kid < 65536:
            name = _lib.MISC_KERNELS.get(kid, "")
            key = "cwct" if name.startswith("cwct") else ("stage3" if name == "presplit" else "glue")
            stage_ms[key] += m / n_frames
            if pmc:
                for pat in MISC_PMC.get(kid, ()):
                    hit = [v for n, v in pmc["kernels"].items() if n.split("(")[0].split("<")[0].replace("void ", "") == pat]
                    if hit:
                        stage_pmc[key] += hit[0]["hbm_bytes_per_launch"] * c / n_frames
    # algorithmic bytes per frame: 192 B per full-res pixel and block, 396 B glue per pass, 384 B cWCT
    blocks = {"stage1": 10 + 10 - 1, "stage2": 20, "stage3": 24}          # forward block 0 is folded into the input packing
    stage_bytes = {k: v * 192.0 * px for k, v in blocks.items()}
    # glue: 396 B per pixel and pass with the spread / gather copies; 140 without them (packed code: only the
    # RGB <-> state boundary kernels remain)
    has_spread = any(_lib.MISC_KERNELS.get(kid, "") in ("spread", "gather") for kid in table if kid < 65536)
    stage_bytes["glue"] = 2 * (396.0 if has_spread else 140.0) * px
    stage_bytes["cwct"] = 384.0 * px
    stages = {}
    for k, m in stage_ms.items():
        stages[k] = {"ms_per_frame": round(m, 4), "algorithmic_TBps": round(stage_bytes[k] / (m * 1e-3) / 1e12, 3) if m > 0 else None}
        if pmc_complete and m > 0:
            stages[k]["pmc_bytes_per_frame"] = int(stage_pmc[k])
            stages[k]["pmc_TBps"] = round(stage_pmc[k] / (m * 1e-3) / 1e12, 3)
        if k in stage_issued:
            stages[k]["issued_PFLOPps"] = round(stage_issued[k] / (m * 1e-3) / 1e15, 3) if m > 0 else None
    stages["sum_ms_per_frame"] = round(sum(stage_ms.values()), 4)
    stages["note"] = "HIP-event kernel time of one frame at a time (launch gaps included per launch); stage3 includes channel_reduction"
    # ---- the frame as a whole --------------------------------------------------------------------------------------------
    passes = 3 if args.recompute_style else 2
    alg_frame = (passes * 6540 + 384) * H * W * fpg + (128 * H * W if args.recompute_style else 0)
    issued_frame = sum(stage_issued.values())
    frame = {"algorithmic_bytes_per_frame": int(alg_frame), "issued_mfma_flops_per_frame": int(issued_frame),
             "kernel_time_sum_ms": stages["sum_ms_per_frame"], "frames_in_flight": frames_in_flight}
    if ms_per_step:
        t = ms_per_step * 1e-3
        frame.update({"ms_per_frame_timed_region": round(ms_per_step, 4),
                      "overlap_of_kernel_time": round(1.0 - ms_per_step / stages["sum_ms_per_frame"], 4),
                      "chip_average_issued_PFLOPps": round(issued_frame / t / 1e15, 3),
                      "chip_average_issued_frac_of_mfma_peak": round(issued_frame / t / 1e12 / MFMA_16BIT_PEAK_TFLOPS, 4),
                      "chip_average_algorithmic_TBps": round(alg_frame / t / 1e12, 3)})
    if pmc_complete and not args.recompute_style:
        pmc_frame = sum(stage_pmc.values())
        frame.update({"pmc_bytes_per_frame": int(pmc_frame), "pmc_over_algorithmic": round(pmc_frame / alg_frame, 3),
                      "pmc_source": traffic_source})
        if ms_per_step:
            frame["chip_average_pmc_TBps"] = round(pmc_frame / (ms_per_step * 1e-3) / 1e12, 3)
            frame["chip_average_pmc_frac_of_hbm_peak"] = round(pmc_frame / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            frame["chip_average_pmc_frac_of_measured_copy_rate"] = round(pmc_frame / (ms_per_step * 1e-3) / 1e9 / HBM_COPY_GBS, 4)
    frame["note"] = ("pmc_bytes_per_frame = sum over this run's launches per frame (HIP-event table) x HBM bytes per launch of the committed "
                     "PMC summary; chip averages divide by the timed region's ms per frame (frames in flight on several streams); "
                     "overlap_of_kernel_time = 1 - ms per frame / sum of kernel times of a frame run alone")
    return roof, stages, [classes[k] for k in top[:5]], frame


def dry_run(args, rank, world):
    """No GPU: the launcher / rendezvous / timing / JSON path only (covered by tests/test_bench_launch.py)."""
    import torch.distributed as dist
    from vstnet_amd.sharding import timed_steps
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    fpg = args.frames_per_gpu
    elapsed, per_rank_s = timed_steps(lambda: time.sleep(args.dry_run_ms * 1e-3), args.steps, args.warmup, lambda: None, world,
                                      return_all=True)
    per_rank_fps = [round(fpg * args.steps / t, 3) for t in per_rank_s]
    rec = {"metric": "stylized frames/sec at 1024x1024 (1/2/4/8 GPU) + % HBM roofline", "value": round(world * fpg * args.steps / elapsed, 3),
           "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "none", "data": "dry-run (no GPU work; not a measurement)",
           "config": {"workload": "dry run", "frames_per_gpu": fpg},
           "per_rank": {"frames_per_s": per_rank_fps, "min": min(per_rank_fps), "max": max(per_rank_fps)}}
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _parity(gpu, ref):
    par = {}
    for name, r in zip(("z_c", "z_cs", "stylized"), ref):
        g = gpu[name].double()
        r = r.double()
        par[name] = {"rel_l2": float(((g - r).norm() / r.norm())), "max_rel": float(((g - r).abs().max() / r.abs().max()))}
    return par


def cpu_baseline(sd, sp, H, W, gpu_out=None, timed_frames=3, mode_out=None, modes_rec=None):
    """The oracle (CPU restatement of the reference's torch-op sequence) timed on this host's cores: one warm-up frame, then
    `timed_frames` frames of the same workload (style code precomputed, like the GPU leg).  The warm-up frame is the GPU
    leg's frame 0, so its code / transferred code / stylised frame are also the parity check of the GPU outputs."""
    import torch
    from oracle import cpu_ref
    from vstnet_amd.synth import synthetic_frames
    cores = usable_cores()
    torch.set_num_threads(cores)
    xs = synthetic_frames(1, H, W, seed=1)
    xc = synthetic_frames(1 + timed_frames, H, W, seed=0)      # frame f = seed (0, f), as in the GPU leg
    with torch.no_grad():
        zs = cpu_ref.revnet_forward(xs, sd, sp)

        def frame(i):
            zc = cpu_ref.revnet_forward(xc[i:i + 1], sd, sp)
            zcs = cpu_ref.transfer(zc, zs)
            return zc, zcs, cpu_ref.revnet_inverse(zcs, sd, sp)
        ref = frame(0)                                          # warm-up (thread pool, allocator) + parity reference
        times = []
        for i in range(1, 1 + timed_frames):
            t0 = time.perf_counter()
            frame(i)
            times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    rec = {"value": round(1.0 / dt, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "threads": torch.get_num_threads(),
           "cpu_model": cpu_model(), "host_cores_visible": os.cpu_count(), "kind": "port",
           "sample": f"1 warm-up + {timed_frames} timed frames {W}x{H} (forward + cWCT + inverse, style code precomputed), "
                     "oracle/cpu_ref.py on torch CPU ops", "seconds_per_frame": [round(t, 2) for t in times]}
    for m, out in (mode_out or {}).items():           # every arithmetic mode's frame 0 against the same oracle frame
        if modes_rec is not None and m in modes_rec:
            par = _parity(out, ref)
            modes_rec[m]["parity_rel_l2"] = float(f"{max(v['rel_l2'] for v in par.values()):.3e}")
            modes_rec[m]["parity_max_rel"] = float(f"{max(v['max_rel'] for v in par.values()):.3e}")
            modes_rec[m]["parity_stylized_rel_l2"] = float(f"{par['stylized']['rel_l2']:.3e}")
    if gpu_out:
        par = _parity(gpu_out, ref)
        rec["parity_rel_l2"] = max(v["rel_l2"] for v in par.values())
        rec["parity_max_rel"] = max(v["max_rel"] for v in par.values())
        rec["parity"] = {k: {a: float(f"{b:.3e}") for a, b in v.items()} for k, v in par.items()}
        rec["parity_note"] = "GPU frame 0 of this run vs the oracle on the same inputs and weights; budget 1e-3 (north_star)"
    return rec


if __name__ == "__main__":
    main()
