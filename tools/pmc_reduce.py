#!/usr/bin/env python3
"""Reduce one `rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d DIR -- python3 <program>` pass into a per-kernel
table: launches, total ms, the sum of every collected counter and the derived figures they allow
(effective clock = GRBM_GUI_ACTIVE / 8 XCDs / wall; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (clock cycles x 1024 SIMDs);
wait / active shares of SQ_WAVE_CYCLES; LDS conflict share).  usage: pmc_reduce.py DIR [out.csv] [name-substring]"""
import collections
import csv
import glob
import re
import sys


def short(name):
    return re.sub(r"\(seg::.*", "", name.replace("void ", "").replace("(anonymous namespace)::", ""))[:80]


def main():
    d = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    sub = sys.argv[3] if len(sys.argv) > 3 else "seg::"
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    assert files, d
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    names = []
    for row in csv.DictReader(open(files[0])):
        k = short(row["Kernel_Name"])
        if sub not in k:
            continue
        cn = row["Counter_Name"]
        if cn not in names:
            names.append(cn)
        acc[k][cn] += float(row["Counter_Value"])
        if row["Dispatch_Id"] not in seen[k]:
            seen[k].add(row["Dispatch_Id"])
            acc[k]["ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    lines = ["kernel,launches,total_ms," + ",".join(names) + ",eff_clock_GHz,mfma_busy,wait_any_share,wait_inst_share,active_inst_share,lds_conflict_share"]
    for k, c in sorted(acc.items(), key=lambda kv: -kv[1]["ns"]):
        sec = c["ns"] * 1e-9
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        clk = cyc / sec / 1e9 if sec > 0 else 0.0
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0) if cyc else 0.0
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        sh = lambda n: (c.get(n, 0.0) / wc) if wc else 0.0
        lds = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else 0.0
        lines.append(f"\"{k}\",{len(seen[k])},{c['ns'] * 1e-6:.3f}," + ",".join(f"{c.get(n, 0.0):.0f}" for n in names) +
                     f",{clk:.3f},{mf:.3f},{sh('SQ_WAIT_ANY'):.3f},{sh('SQ_WAIT_INST_ANY'):.3f},{sh('SQ_ACTIVE_INST_ANY'):.3f},{lds:.3f}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
