"""Summarise one `rocprofv3 --kernel-trace --pmc <SQ counters>` pass per kernel (mean per launch):

  python tools/pmc_sq_summary.py gpurun_out/pmc_sq profiles/r03_pmc_sq.json "<command>" "<workload>" [precision] [git commit]

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD
summed over SIMDs (MI355X_MICROARCH.md, cycle constants).  Reported per kernel: the share of wave time parked (s_waitcnt /
barrier), stalled at issue, issuing; MFMA-busy cycles per SIMD against the kernel's duration in shader cycles (GRBM_GUI_ACTIVE / 8
XCDs); LDS-array activity and bank-conflict cycles per CU."""
import collections
import csv
import glob
import json
import os
import sys


def main(prof_dir, dest, command, workload, precision=None, git_commit=None):
    f = max(glob.glob(os.path.join(prof_dir, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
    out = {}
    for k, c in tot.items():
        n = len(disp[k])
        wave = c.get("SQ_WAVE_CYCLES", 0.0)
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / n            # shader cycles of one launch (sum over the 8 XCDs / 8)
        rec = {"launches": n}
        if wave:
            rec.update(wait_any_frac=round(c.get("SQ_WAIT_ANY", 0) / wave, 3), wait_inst_any_frac=round(c.get("SQ_WAIT_INST_ANY", 0) / wave, 3),
                       active_inst_any_frac=round(c.get("SQ_ACTIVE_INST_ANY", 0) / wave, 3),
                       wait_inst_lds_frac=round(c.get("SQ_WAIT_INST_LDS", 0) / wave, 3))
        if gui:
            rec["kernel_cycles"] = int(gui)
            rec["mfma_busy_frac_per_simd"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / n / 1024.0 / gui, 3)    # 256 CUs x 4 SIMDs
            rec["lds_active_frac_per_cu"] = round(c.get("SQ_LDS_IDX_ACTIVE", 0) / n / 256.0 / gui, 3)
            rec["lds_bank_conflict_frac_per_cu"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / n / 256.0 / gui, 3)
        out[k] = rec
    json.dump({"command": command, "workload": workload, "precision": precision, "git_commit": git_commit,
               "units": "fractions of wave time (SQ_WAVE_CYCLES) / of the launch's shader cycles (GRBM_GUI_ACTIVE / 8)",
               "kernels": out}, open(dest, "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("kernel_cycles", 0) * kv[1]["launches"])[:12]:
        print(f"{k[:64]:64s} n={v['launches']:4d} cyc={v.get('kernel_cycles', 0):7d} mfma={v.get('mfma_busy_frac_per_simd')} "
              f"wait={v.get('wait_any_frac')} stall={v.get('wait_inst_any_frac')} issue={v.get('active_inst_any_frac')} "
              f"lds={v.get('lds_active_frac_per_cu')} conf={v.get('lds_bank_conflict_frac_per_cu')}")


if __name__ == "__main__":
    main(*sys.argv[1:7])
