"""Overlapped host <-> device frame loop for the video path (replaces the synchronous
per-frame loop of video_transfer.py:186-214).

Frames enter and leave as uint8 HWC in PINNED host ring buffers.  Each frame's H2D copy, encoder pass, cWCT, decoder
pass and D2H copy are queued on one of `compute_streams` HIP streams (consecutive frames alternate, so that the
copies and kernels of two frames overlap on the card); the host only waits when it retires frame i-(depth-1), i.e.
after it has queued `depth-1` newer frames.  A uint8 frame is ~0.1 ms of PCIe time against milliseconds of compute, so
dedicated copy streams (and the cross-stream events they need) would buy nothing.  Nothing here touches
pixels: the output is bit-identical to `net.inverse_u8(transform(net.forward_u8(frame)))` run one frame at a time
(tests/test_gpu_parity.py::test_frame_pipeline_matches_sequential).

`prefetch()` runs a frame source (decode + resize) in a background thread; PIL and numpy release the GIL in their
inner loops, so decode, the GPU and the sink (encode) overlap.
"""
from __future__ import annotations

import queue
import threading

import numpy as np
import torch


class FramePipeline:
    def __init__(self, net, transform, height, width, device=None, depth=4, compute_streams=2, decode=None,
                 out_height=None, out_width=None):
        """net: vstnet_amd RevResNet on the GPU; transform(z_c, index) -> z_cs runs on the current stream (cWCT);
        height/width: the (fixed) frame size, multiples of 4; depth: ring slots (>= 2).  decode(z_cs) -> uint8
        [1,out_height,out_width,3] device tensor replaces net.inverse_u8 when the written size differs from the
        stylised size (the reference's writer-size quirk, video_transfer.py:83-86,210-212)."""
        if not torch.cuda.is_available():
            raise RuntimeError("FramePipeline needs the GPU (no CPU fallback)")
        if depth < 2:
            raise ValueError("depth must be >= 2")
        if height % 4 or width % 4 or height < 8 or width < 8:
            raise ValueError(f"frame size must be multiples of 4 and >= 8 (got {height}x{width})")
        self.net, self.transform = net, transform
        self.decode = decode if decode is not None else net.inverse_u8
        self.H, self.W, self.depth = height, width, depth
        self.Ho, self.Wo = out_height or height, out_width or width
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        with torch.cuda.device(self.device):
            self.h_in = torch.empty((depth, height, width, 3), dtype=torch.uint8).pin_memory()
            self.h_out = torch.empty((depth, self.Ho, self.Wo, 3), dtype=torch.uint8).pin_memory()
            self.h_in_np, self.h_out_np = self.h_in.numpy(), self.h_out.numpy()
            self.d_in = torch.empty((depth, 1, height, width, 3), dtype=torch.uint8, device=self.device)
            self.s_comp = [torch.cuda.Stream(device=self.device) for _ in range(max(1, compute_streams))]
            # fp16 modes: the library's range flags travel with every frame (4 words, appended to its D2H copy) and are looked
            # at when the frame is retired - saturation is an error of THAT frame, not a silent clamp somewhere in the clip
            self.d_flags = torch.zeros((depth, 4), dtype=torch.int32, device=self.device)
            self.h_flags = torch.zeros((depth, 4), dtype=torch.int32).pin_memory()
            self.h_flags_np = self.h_flags.numpy()
            self.flag_check = [False] * depth
            self.done = [torch.cuda.Event() for _ in range(depth)]
            self.consumed = [torch.cuda.Event() for _ in range(depth)]      # compute(i) has read d_in[slot]

    def _submit(self, i, frame):
        k = i % self.depth
        src = frame.numpy() if isinstance(frame, torch.Tensor) else np.asarray(frame)
        if src.shape != (self.H, self.W, 3) or src.dtype != np.uint8:
            raise ValueError(f"frame {i}: expected uint8 [{self.H},{self.W},3], got {src.dtype} {tuple(src.shape)}")
        # plain single-threaded memcpy into the pinned slot.  (Not torch's CPU copy_: its intra-op thread pool spins after
        # every call and, inside a CPU-quota cgroup, throttles the thread that feeds the GPU — measured 9 ms vs 0.3 ms.)
        np.copyto(self.h_in_np[k], src)
        sc = self.s_comp[i % len(self.s_comp)]
        with torch.cuda.device(self.device), torch.no_grad(), torch.cuda.stream(sc):
            # H2D, compute and D2H of one frame are queued on ONE stream (a uint8 frame is ~0.1 ms of PCIe time against
            # milliseconds of compute); overlap comes from consecutive frames being on different streams
            if i >= self.depth:
                sc.wait_event(self.consumed[k])                   # slot reuse: the previous tenant's encoder pass has read it
            self.d_in[k].copy_(self.h_in[k].unsqueeze(0), non_blocking=True)
            z_c = self.net.forward_u8(self.d_in[k])
            self.consumed[k].record(sc)
            out = self.decode(self.transform(z_c, i))
            if tuple(out.shape) != (1, self.Ho, self.Wo, 3) or out.dtype != torch.uint8:
                raise RuntimeError(f"decode returned {out.dtype} {tuple(out.shape)}, expected uint8 (1,{self.Ho},{self.Wo},3)")
            self.h_out[k].copy_(out[0], non_blocking=True)
            self.flag_check[k] = getattr(self.net, "resolved_precision", None) in ("f16x2", "f16x2h")
            if self.flag_check[k]:
                from . import _lib
                import ctypes as C
                _lib.check(_lib.lib().vst_range_flags_async(C.c_void_p(self.d_flags[k].data_ptr()), C.c_void_p(sc.cuda_stream)),
                           "vst_range_flags_async")
                self.h_flags[k].copy_(self.d_flags[k], non_blocking=True)
            self.done[k].record(sc)

    def _retire(self, i, sink):
        k = i % self.depth
        self.done[k].synchronize()
        if self.flag_check[k] and self.h_flags_np[k].any():
            flags = int(np.bitwise_or.reduce(self.h_flags_np[k]))
            raise RuntimeError(f"frame {i}: fp16 range flags 0x{flags:x} raised by precision='{self.net.resolved_precision}' "
                               "(1 = an activation saturated at +-65504): this checkpoint / input needs precision='bf16x3'")
        sink(i, self.h_out_np[k])        # a view of the pinned slot: valid until `depth` more frames are submitted

    def run(self, frames, sink, start_index=0):
        """frames: iterable of uint8 HWC arrays/tensors; sink(index, uint8 HWC numpy view) is called in frame order
        from this thread (copy or encode before returning).  Returns the number of frames processed."""
        n = 0
        lag = self.depth - 1
        with torch.cuda.device(self.device):       # whatever the caller queued so far (style code, statistics) comes first
            ev0 = torch.cuda.Event()
            ev0.record(torch.cuda.current_stream())
            for st in self.s_comp:
                st.wait_event(ev0)
        for frame in frames:
            # slot (n % depth) was last used by frame n-depth, which was retired in the previous iteration
            self._submit(start_index + n, frame)
            n += 1
            if n > lag:
                self._retire(start_index + n - 1 - lag, sink)
        for j in range(max(0, n - lag), n):
            self._retire(start_index + j, sink)
        return n


def prefetch(source, ahead=4):
    """Iterate `source` in a background thread, up to `ahead` items ahead of the consumer; exceptions are re-raised
    in the consumer."""
    q: queue.Queue = queue.Queue(maxsize=max(1, ahead))
    end = object()
    stop = threading.Event()            # set when the consumer stops early (exception, break, generator close)

    def put(item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def work():
        try:
            for item in source:
                if not put(item):
                    return              # consumer is gone: drop the decoded frames, end the thread
            put(end)
        except BaseException as e:      # noqa: BLE001 — handed to the consumer
            put(e)

    t = threading.Thread(target=work, daemon=True)
    t.start()
    try:
        while True:
            item = q.get()
            if item is end:
                break
            if isinstance(item, BaseException):
                raise item
            yield item
    finally:
        stop.set()
        while True:                     # release whatever the producer had queued
            try:
                q.get_nowait()
            except queue.Empty:
                break
        t.join(timeout=5.0)


def parallel_map(fn, items, workers=4, ahead=8):
    """fn(item) for every item on `workers` threads, results yielded IN ORDER, at most `ahead` + `workers` items in flight
    (bounded memory: a decoded 1080p frame is 6 MB).  PIL's decoders, resizers and numpy copies release the GIL, so the frames
    of a clip decode on several cores while one thread feeds the GPU.  An exception of fn is re-raised at its item's turn."""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    if workers <= 1:
        for item in items:
            yield fn(item)
        return
    pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="vst-decode")
    pending = deque()
    try:
        it = iter(items)
        for item in it:
            pending.append(pool.submit(fn, item))
            if len(pending) >= ahead + workers:
                yield pending.popleft().result()
        while pending:
            yield pending.popleft().result()
    finally:
        for f in pending:
            f.cancel()
        pool.shutdown(wait=True)


def host_cores():
    """cores this process may use: the usable cores (affinity mask, cgroup quota), capped by the per-rank thread cap that
    launch_children sets in OMP_NUM_THREADS"""
    import os
    from .sharding import usable_cores
    n = usable_cores()
    if os.environ.get("OMP_NUM_THREADS", "").isdigit():
        n = min(n, int(os.environ["OMP_NUM_THREADS"]))
    return max(1, n)


def host_workers(requested=0):
    """(decode threads, encode threads) of the video loop: `requested` of each, or a split of this process's cores that leaves
    one for the thread that feeds the GPU - a frame's PNG encode costs several times its decode + resize"""
    if requested > 0:
        return requested, requested
    n = host_cores() - 1
    dec = max(1, min(4, n // 4))
    return dec, max(1, min(12, n - dec))


def save_png(path, frame, level=0):
    """uint8 [H,W,3] -> a PNG file.  level 0: this module's own writer - unfiltered rows in stored deflate blocks (a valid,
    lossless PNG of 3 bytes per pixel + 0.02 %): a few milliseconds per 1080p frame where PIL's encoder spends 50-140 ms on row
    filters even at compress_level 0; the zlib / crc32 calls release the GIL, so several sink threads scale.  level 1-9: PIL with
    that zlib level (filters + deflate: ~30 % smaller files on photographic content, 10-30x the time)."""
    import struct
    import zlib
    arr = np.ascontiguousarray(frame, dtype=np.uint8)
    if level > 0 or arr.ndim != 3 or arr.shape[2] != 3:
        from PIL import Image
        Image.fromarray(arr).save(path, format="PNG", compress_level=max(0, level))
        return
    h, w, _ = arr.shape
    raw = np.empty((h, 1 + 3 * w), np.uint8)
    raw[:, 0] = 0                                           # filter type 0 (none) in front of every row
    raw[:, 1:] = arr.reshape(h, 3 * w)
    data = zlib.compress(raw, 0)

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(body, zlib.crc32(tag)) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        f.write(struct.pack(">I", len(data)) + b"IDAT")
        f.write(data)
        f.write(struct.pack(">I", zlib.crc32(data, zlib.crc32(b"IDAT")) & 0xFFFFFFFF) + chunk(b"IEND", b""))


class AsyncSink:
    """Run a sink (frame encode / file write) on background threads; frames are copied out of the pinned slot first.
    workers = 1: one thread, frames reach `fn` in order (a video writer).  workers > 1: `fn` is called concurrently and in any
    order - for sinks whose calls are independent (one numbered PNG per frame: the encode of a 1080p frame is tens of
    milliseconds on one core, several times the GPU time of the frame).  close() waits for the queue to drain and re-raises the
    first writer error."""

    def __init__(self, fn, ahead=8, workers=1):
        self.fn = fn
        self.q: queue.Queue = queue.Queue(maxsize=max(1, ahead, 2 * workers))
        self.err = None
        self.threads = [threading.Thread(target=self._work, daemon=True, name=f"vst-sink-{k}") for k in range(max(1, workers))]
        for t in self.threads:
            t.start()

    def _work(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if self.err is None:
                try:
                    self.fn(*item)
                except BaseException as e:      # noqa: BLE001
                    self.err = e

    def __call__(self, index, frame):
        if self.err is not None:
            raise self.err
        self.q.put((index, np.array(frame, copy=True)))

    def close(self):
        for _ in self.threads:
            self.q.put(None)
        for t in self.threads:
            t.join()
        if self.err is not None:
            raise self.err
