#!/usr/bin/env python3
"""Reduce one rocprofv3 --pmc pass (SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE) of
`bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-exact-leg` into a per-kernel CSV:
launches, total ms, effective clock (GRBM_GUI_ACTIVE / 8 XCDs / wall: MI355X_MICROARCH.md "DVFS give-back"), MFMA busy as a
fraction of the chip's 1024 SIMDs, LDS bank-conflict cycles over active LDS cycles.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
            -d $R/gpurun_out/pmc_sq -o s -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-exact-leg
  python tools/pmc_sq_summary.py gpurun_out/pmc_sq profiles/r02_pmc_sq_summary.csv"""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    return name[:90]


def main():
    d, out = sys.argv[1:3]
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    assert files, d
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for row in csv.DictReader(open(files[0])):
        k = short(row["Kernel_Name"])
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Dispatch_Id"] not in seen[k]:
            seen[k].add(row["Dispatch_Id"])
            acc[k]["ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    rows = []
    for k, c in acc.items():
        if "seg::" not in k or c["ns"] <= 0:
            continue
        sec = c["ns"] * 1e-9
        clk = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / sec / 1e9
        mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c.get("GRBM_GUI_ACTIVE", 1.0) / 8.0 * 1024.0) if c.get("GRBM_GUI_ACTIVE") else 0.0
        lds = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else 0.0
        rows.append((c["ns"], k, len(seen[k]), c["ns"] * 1e-6, clk, mfma, lds))
    rows.sort(reverse=True)
    with open(out, "w") as fh:
        fh.write("kernel,launches,total_ms,eff_clock_GHz,mfma_busy_frac_of_1024_SIMDs,lds_conflict_over_active\n")
        for _, k, n, ms, clk, mfma, lds in rows:
            fh.write(f"\"{k}\",{n},{ms:.3f},{clk:.3f},{mfma:.3f},{lds:.3f}\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
