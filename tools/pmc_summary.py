"""Combine two `rocprofv3 --pmc` passes (FETCH_SIZE and WRITE_SIZE, each with --kernel-trace --output-format csv) into
the per-kernel HBM-traffic summary kept under profiles/ (MI355X_MICROARCH.md, HBM section: FETCH_SIZE is in KB and
counts 64 B per 128-B request on gfx950 wide coalesced reads -> doubled; WRITE_SIZE in KB, exact).

  python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r03_pmc_hbm_traffic.json "<command>" "<workload>" \
      [precision] [git commit]      (the last two are what bench.py's roofline.traffic_source reports)
"""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(prof_dir, counter):
    f = max(glob.glob(os.path.join(prof_dir, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    tot, disp = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: (tot[k], len(disp[k])) for k in tot}


def main(fetch_dir, write_dir, dest, command, workload, precision=None, git_commit=None):
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    kernels = {}
    for k, (fkb, n) in fetch.items():
        wkb, nw = write.get(k, (0.0, n))
        fb, wb = fkb * 1024.0 * 2.0 / n, wkb * 1024.0 / max(nw, 1)
        kernels[k] = {"launches": n, "fetch_size_kb_raw": round(fkb / n, 1), "fetch_bytes_corrected": int(fb),
                      "write_bytes": int(wb), "hbm_bytes_per_launch": int(fb + wb)}
    json.dump({"command": command,
               "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> doubled "
                             "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact. Both are L2 memory-side request "
                             "counters: Infinity-Cache hits are included.",
               "workload": workload, "precision": precision, "git_commit": git_commit, "kernels": kernels},
              open(dest, "w"), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:14]:
        print(f"{k[:70]:70s} n={v['launches']:4d} {v['hbm_bytes_per_launch'] / 2 ** 20:8.1f} MiB/launch")


if __name__ == "__main__":
    main(*sys.argv[1:8])
