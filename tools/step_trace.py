#!/usr/bin/env python3
"""Per-launch timeline of the last of N identical steps from a rocprofv3 kernel trace: step_trace.py kernel_trace.csv N [marker]
(launch order, duration, idle gap in front of each launch; the trace is cut into N equal runs of launches from its end)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the period: the launch count of one step = distance between the last two launches of the (once-per-step) loss kernel
marker = sys.argv[3] if len(sys.argv) > 3 else "bce_argmax_dice_kernel"      # a kernel launched once per step
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
per = marks[-1] - marks[-2]
last = rows[marks[-2] + 1: marks[-1] + 1]
# rotate so that the step starts at its first kernel after the optimizer (the fused Adam launches end a step)
t_prev = int(rows[marks[-2]]["End_Timestamp"])
tot = gap = 0.0
print(f"# {per} launches per step")
agg = {}
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = re.sub(r"^void ", "", r["Kernel_Name"])
    nm = re.sub(r"\(anonymous namespace\)::", "", nm)[:100]
    d, g = (e - s) / 1e3, (s - t_prev) / 1e3
    print(f"{d:9.1f} us  gap {g:7.1f}  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>9} lds {r.get('LDS_Block_Size', '?'):>6} vgpr {r.get('VGPR_Count', '?'):>4}  {nm}")
    tot += d; gap += max(g, 0.0); t_prev = e
    a = agg.setdefault(nm.split("(")[0][:80], [0, 0.0]); a[0] += 1; a[1] += d
print(f"# kernel time {tot / 1e3:.3f} ms, idle gaps {gap / 1e3:.3f} ms, span {(tot + gap) / 1e3:.3f} ms")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"## {t / 1e3:8.3f} ms {c:4d}  {k}")
