"""Turn a `rocprofv3 --kernel-trace --stats` output directory into the table kept under profiles/.

  python tools/profile_summary.py gpurun_out/prof profiles/r01_bench_1024_photo_kernel_stats.csv [profiles/r01_pmc_hbm_traffic.json]

copies the newest *_kernel_stats.csv to the given path and prints a markdown table (kernel, calls, avg us, share,
and HBM MiB per launch when the PMC summary is given)."""
import csv
import glob
import json
import os
import shutil
import sys


def main(prof_dir, dest_csv, pmc_json=None):
    src = max(glob.glob(os.path.join(prof_dir, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    shutil.copyfile(src, dest_csv)
    rows = list(csv.DictReader(open(dest_csv)))
    pmc = json.load(open(pmc_json))["kernels"] if pmc_json else {}
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    print("| kernel | calls | avg us | % of kernel time | HBM MiB/launch (PMC) |")
    print("|---|---|---|---|---|")
    for r in rows:
        if float(r["Percentage"]) < 0.4:
            continue
        name = r["Name"]
        mib = next((v["hbm_bytes_per_launch"] / 2 ** 20 for k, v in pmc.items() if k[:60] == name[:60]), None)
        print(f"| `{name[:80]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} | "
              f"{'%.1f' % mib if mib is not None else '-'} |")
    print(f"\nTotal kernel time {total / 1e6:.1f} ms over the run.")


if __name__ == "__main__":
    main(*sys.argv[1:4])
