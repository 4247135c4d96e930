"""Overlap analysis of a rocprofv3 --kernel-trace CSV of frames in flight on several streams.

    python tools/overlap_trace.py gpurun_out/kt/<host>/<pid>_kernel_trace.csv [--skip-ms 30]

Classes: S3 = the 256-channel convs (conv_pipe / conv_sp kernels: MFMA-bound), S12 = every other conv kernel (HBM-bound),
other = cWCT / layout.  Reports, over the steady part of the trace: the union of busy time, per-class kernel-time sums, and
for every S3 dispatch the fraction of its interval during which an S12 dispatch of ANOTHER queue was also running (and the
reverse) - i.e. whether the two families' intervals overlap at all.  (Same-CU residency is not visible here: that needs the
per-workgroup HW_ID stamps of a -DVST_TRACE build.)
"""
import csv
import json
import sys


def cls(name):
    if "conv_pipe_kernel" in name or "conv_sp_kernel" in name:
        return "S3"
    if "conv_mfma_kernel" in name or "conv_pair_kernel" in name:
        return "S12"
    return "other"


def main():
    path = sys.argv[1]
    skip_ms = float(sys.argv[sys.argv.index("--skip-ms") + 1]) if "--skip-ms" in sys.argv else 0.0
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), cls(r["Kernel_Name"]), r["Kernel_Name"]))
    rows.sort()
    # steady part: from the first S3 dispatch after which >= 2 queues are active, to the last dispatch start
    conv = [r for r in rows if r[3] != "other"]
    t_first = conv[0][0]
    # drop warm-up (style encode etc. on the default stream): keep dispatches of the non-default queues
    qcount = {}
    for r in conv:
        qcount[r[2]] = qcount.get(r[2], 0) + 1
    t0 = t_first + int(skip_ms * 1e6)
    t1 = max(r[1] for r in rows)
    sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
    # trim the tail where fewer streams are active: stop at the start of the last frame's first kernel is hard; use 90 %
    t1 = t0 + int(0.9 * (t1 - t0))
    sel = [r for r in sel if r[1] <= t1]
    span = (t1 - t0) / 1e6
    sums = {"S3": 0.0, "S12": 0.0, "other": 0.0}
    for s, e, q, c, n in sel:
        sums[c] += (e - s) / 1e6
    # union of busy time
    ev = sorted([(s, 1) for s, e, q, c, n in sel] + [(e, -1) for s, e, q, c, n in sel])
    busy, depth, last = 0, 0, None
    hist = {}
    for t, d in ev:
        if depth > 0:
            busy += t - last
        if last is not None:
            hist[depth] = hist.get(depth, 0) + (t - last)
        depth += d
        last = t
    # pairwise overlap S3 x S12 across queues
    s3 = [r for r in sel if r[3] == "S3"]
    s12 = [r for r in sel if r[3] == "S12"]
    import bisect
    s12_sorted = sorted(s12)
    starts = [r[0] for r in s12_sorted]
    ov_total = 0
    for s, e, q, c, n in s3:
        i = bisect.bisect_left(starts, s - 200000)
        while i < len(s12_sorted) and s12_sorted[i][0] < e:
            ss, ee, qq = s12_sorted[i][:3]
            if qq != q:
                ov_total += max(0, min(e, ee) - max(s, ss))
            i += 1
    # same-class overlap (S3 with S3 of another queue)
    s3s = sorted(s3)
    st3 = [r[0] for r in s3s]
    ov33 = 0
    for s, e, q, c, n in s3:
        i = bisect.bisect_left(st3, s - 200000)
        while i < len(s3s) and s3s[i][0] < e:
            ss, ee, qq = s3s[i][:3]
            if qq != q and (ss, ee, qq) != (s, e, q):
                ov33 += max(0, min(e, ee) - max(s, ss))
            i += 1
    # average duration per kernel name
    per = {}
    for s, e, q, c, n in sel:
        k = n.split("(")[0][:70]
        a = per.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    rec = {"file": path, "span_ms": round(span, 3), "busy_union_ms": round(busy / 1e6, 3), "queues": qcount,
           "kernel_time_sum_ms": {k: round(v, 3) for k, v in sums.items()},
           "sum_over_span": round(sum(sums.values()) / span, 3),
           "depth_hist_ms": {str(k): round(v / 1e6, 3) for k, v in sorted(hist.items())},
           "S3_time_overlapped_by_other_queue_S12_ms": round(ov_total / 1e6, 3),
           "S3_overlap_frac": round(ov_total / 1e6 / max(sums["S3"], 1e-9), 3),
           "S3_x_S3_other_queue_ms": round(ov33 / 2 / 1e6, 3)}
    print(json.dumps(rec, indent=1))
    for k, (cnt, us) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"{us / cnt:9.1f} us x {cnt:5d}  {k}")


if __name__ == "__main__":
    main()
