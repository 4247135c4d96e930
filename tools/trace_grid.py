"""Whole-grid timeline of one stage-1/2 launch (diagnostic builds only).

    hipcc ... -DVST_TRACE=1 (conv_pair_kernel) or =2 (conv_mfma_kernel) -o vstnet_amd/abl/traceN.so vstnet_amd/csrc/*.hip
    VSTNET_HIP_LIB=$PWD/vstnet_amd/abl/trace1.so python tools/trace_grid.py --block 64:1

Every workgroup of the last traced launch leaves {start, end, HW_ID, XCC_ID}; this prints how many workgroups each CU ran,
how many were resident together, how long a workgroup lived early and late in the launch, and the idle share of the CUs.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--block", default="64:1")
    ap.add_argument("--precision", default="f16x2")
    args = ap.parse_args()
    L = _lib.lib()
    L.vst_trace_dump.restype = C.c_int
    L.vst_trace_dump.argtypes = [C.c_void_p, C.c_int]
    H = W = args.size
    dev = torch.device("cuda", 0)
    net = RevResNet().to(dev).eval()
    net.load_state_dict(synthetic_state_dict())
    w = net._ensure_packed(dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ch, stride = (int(v) for v in args.block.split(":"))
    div = {16: 1, 64: 2, 256: 4}[ch]
    kidx = {(16, 1): 3, (64, 1): 13, (64, 2): 10, (256, 1): 25, (256, 2): 20}[(ch, stride)]
    dst = torch.randn(1, H // div, W // div, ch, device=dev)
    src = torch.randn(1, H // div, W // div, ch, device=dev)
    tmp = torch.empty(L.vst_block_tmp_bytes(1, H, W), dtype=torch.uint8, device=dev)
    prec = {"bf16x3": _lib.PREC_BF16X3, "f16x2": _lib.PREC_F16X2}[args.precision]
    for _ in range(6):
        _lib.check(L.vst_block_apply(C.byref(w.blocks[kidx]), ch, stride, 1, prec, C.c_void_p(dst.data_ptr()),
                                     C.c_void_p(src.data_ptr()), C.c_void_p(tmp.data_ptr()), 1, H, W, st), "block")
    torch.cuda.synchronize()
    n_wg = ((H // div + 15) // 16) * ((W // div + 15) // 16)
    n_wg = (n_wg + 7) // 8 * 8
    buf = np.zeros((n_wg, 4), dtype=np.uint64)
    rc = L.vst_trace_dump(buf.ctypes.data_as(C.c_void_p), n_wg)
    assert rc == 0, rc
    buf = buf[buf[:, 1] > 0]
    t0, t1 = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64)
    base = t0.min()
    t0, t1 = (t0 - base) * 0.01, (t1 - base) * 0.01              # microseconds
    hw, xcc = buf[:, 2].astype(np.int64), buf[:, 3].astype(np.int64) & 15
    cu = (hw >> 8) & 15
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    span = t1.max()
    dur = t1 - t0
    ids = np.unique(cuid)
    print(f"{len(buf)} workgroups on {len(ids)} CUs; launch span {span:.1f} us; workgroup life mean {dur.mean():.2f} us "
          f"(min {dur.min():.2f}, median {np.median(dur):.2f}, max {dur.max():.2f})")
    per_cu = np.array([np.sum(cuid == i) for i in ids])
    print(f"workgroups per CU: min {per_cu.min()}, mean {per_cu.mean():.2f}, max {per_cu.max()};  "
          f"per XCD: {[int(np.sum(xcc == x)) for x in range(8)]}")
    # residency: average number of live workgroups per CU over the span, and the share of CU-time with none
    grid = np.linspace(0, span, 400, endpoint=False)
    live = np.zeros((len(ids), len(grid)), dtype=np.int32)
    for k, i in enumerate(ids):
        m = cuid == i
        live[k] = ((t0[m][:, None] <= grid[None, :]) & (t1[m][:, None] > grid[None, :])).sum(0)
    print(f"live workgroups per CU: mean {live.mean():.2f}, max {live.max()};  CU-time with none live: {np.mean(live == 0) * 100:.1f} %")
    q = np.linspace(0, span, 11)
    print("time slice (us)      live WGs/CU   starts   mean life of WGs started in the slice")
    for a_, b_ in zip(q[:-1], q[1:]):
        sel = (grid >= a_) & (grid < b_)
        st_ = (t0 >= a_) & (t0 < b_)
        print(f"  {a_:6.1f} - {b_:6.1f}      {live[:, sel].mean():5.2f}      {int(st_.sum()):5d}    "
              f"{dur[st_].mean() if st_.any() else float('nan'):6.2f}")
    last = np.array([t1[cuid == i].max() for i in ids])
    print(f"CU finish times: min {last.min():.1f}, median {np.median(last):.1f}, max {last.max():.1f} us")
    first = np.array([t0[cuid == i].min() for i in ids])
    print(f"CU first start: min {first.min():.2f}, median {np.median(first):.2f}, max {first.max():.2f} us")


if __name__ == "__main__":
    main()
