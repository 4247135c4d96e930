"""In-kernel timeline of conv_pp_kernel (diagnostic -DVST_PP_STAMP=1 build): s_memtime stamps of wave 0 (group X) and wave 4
(group Y) of one mid-grid workgroup around every stage's burst and staging segment.

    python tools/ab_build.py vstnet_amd/abl/ppstamp.so -DVST_PP_STAMP=1
    VSTNET_HIP_LIB=$PWD/vstnet_amd/abl/ppstamp.so python tools/pp_stamps.py [--conv 256:64|64:64|64:256]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstnet_amd import _lib                                     # noqa: E402
from vstnet_amd.synth import synthetic_state_dict               # noqa: E402
from models.RevResNet import RevResNet                          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--pingpong", type=int, default=1)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
L = _lib.lib()
L.vst_pp_stamps_dump.restype = C.c_int
L.vst_pp_stamps_dump.argtypes = [C.c_void_p]
if args.pingpong:
    _lib.set_option(_lib.OPT_STAGE3_PINGPONG, 1)          # (needs a -DVST_WITH_PINGPONG=1 build)
L.vst_pp_clk_dump.restype = C.c_int
L.vst_pp_clk_dump.argtypes = [C.c_void_p]
H = W = args.size
dev = torch.device("cuda", 0)
net = RevResNet().to(dev).eval()
net.load_state_dict(synthetic_state_dict())
w = net._ensure_packed(dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
dst = torch.randn(1, H // 4, W // 4, 256, device=dev)
src = torch.randn(1, H // 4, W // 4, 256, device=dev)
tmp = torch.empty(L.vst_block_tmp_bytes(1, H, W), dtype=torch.uint8, device=dev)
# the three convs of a 256-channel block run one after the other; the stamps of the LAST launch (conv.7) remain.  To see the
# others, the block is cut short through the profile of a 2-conv / 1-conv variant: simplest is to look at conv.7 here and at
# conv.1 through VST_PP_STAMP_CONV (compile-time) - this tool prints whatever the build stamped.
for _ in range(args.reps):
    _lib.check(L.vst_block_apply(C.byref(w.blocks[25]), 256, 1, 1, _lib.PREC_BF16X3, C.c_void_p(dst.data_ptr()),
                                 C.c_void_p(src.data_ptr()), C.c_void_p(tmp.data_ptr()), 1, H, W, st), "block")
torch.cuda.synchronize()
clk = np.zeros(4, dtype=np.uint64)
assert L.vst_pp_clk_dump(clk.ctypes.data_as(C.c_void_p)) == 0
clk = clk.astype(np.int64)
cyc, ticks = clk[2] - clk[0], clk[3] - clk[1]
print(f"{os.environ.get('VSTNET_HIP_LIB', 'shipped library').split('/')[-1]}: last stage-3 launch (conv.7), pingpong={args.pingpong}, after {args.reps} blocks back to back: workgroup 128 took {cyc} shader cycles in "
      f"{ticks * 10} ns -> in-kernel clock {cyc / max(ticks, 1) * 0.1:.3f} GHz")
if not args.pingpong:
    sys.exit(0)
buf = np.zeros((2, 32, 8), dtype=np.uint64)
assert L.vst_pp_stamps_dump(buf.ctypes.data_as(C.c_void_p)) == 0
b = buf.astype(np.int64)
t0 = b[0, 0, 0]
print("stage |  X: burst  wait1  segment  wait2 |  Y: land(+prefetch)  rest-of-segment  wait1  burst  wait2   (cycles)")
for s in range(24):
    x, y = b[0, s], b[1, s]
    if x[0] == 0:
        break
    xn = b[0, s + 1, 0] if s + 1 < 24 and b[0, s + 1, 0] else x[3]
    yn = b[1, s + 1, 0] if s + 1 < 24 and b[1, s + 1, 0] else y[4]
    print(f"{s:5d} | {x[1] - x[0]:8d} {x[2] - x[1]:6d} {x[3] - x[2]:8d} {xn - x[3]:6d} | {y[1] - y[0]:10d} {y[2] - y[1]:16d} {y[3] - y[2]:6d} {y[4] - y[3]:6d} {yn - y[4]:6d}"
          f"   X starts at {x[0] - t0}, Y segment at {y[0] - t0}")
print("total X:", b[0, 23, 3] - b[0, 0, 0] if b[0, 23, 3] else "-")
