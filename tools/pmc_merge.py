#!/usr/bin/env python3
"""Merge the two SQ counter passes of `tools/profile_round.sh sq` (sqa_<n>.csv, sqb_<n>.csv from tools/pmc_reduce.py) into the
per-kernel table kept under profiles/: usage  pmc_merge.py sqa.csv sqb.csv out.csv [git-stamp]"""
import csv, sys
a = {r["kernel"]: r for r in csv.DictReader(open(sys.argv[1]))}
b = {r["kernel"]: r for r in csv.DictReader(open(sys.argv[2]))}
stamp = sys.argv[4] if len(sys.argv) > 4 else "?"
with open(sys.argv[3], "w") as f:
    f.write(f"# rocprofv3 --pmc, two SQ passes (tools/profile_round.sh sq) at git {stamp}; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs); shares are of SQ_WAVE_CYCLES\n")
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "total_ms", "eff_clock_GHz", "mfma_busy", "lds_conflict_over_active", "wait_any_share", "wait_inst_share", "active_inst_share", "valu_insts", "lds_insts"])
    for k, r in a.items():
        q = b.get(k, {})
        w.writerow([k, r["launches"], r["total_ms"], r["eff_clock_GHz"], r["mfma_busy"], r["lds_conflict_share"], q.get("wait_any_share", ""), q.get("wait_inst_share", ""),
                    q.get("active_inst_share", ""), r["SQ_INSTS_VALU"], r["SQ_INSTS_LDS"]])
