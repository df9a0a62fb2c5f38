#!/usr/bin/env python3
"""Per-kernel L2-to-fabric traffic from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh (fetch_<w>, write_<w> under
the given directory): launches, bytes per launch (FETCH_SIZE doubled: gfx950 tallies 128-byte requests at 64 bytes, see
MI355X_MICROARCH.md HBM section; WRITE_SIZE as read; both counters are in KB), written as <dir>/traffic_<w>.csv."""
import collections, csv, glob, os, re, sys


def short(name):
    return re.sub(r"\(seg::.*", "", name.replace("void ", "").replace("(anonymous namespace)::", ""))[:90]


def per_kernel(directory, counter):
    files = glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
    if not files:
        return acc
    seen = set()
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != counter or "seg::" not in row["Kernel_Name"]:
            continue
        k = short(row["Kernel_Name"])
        acc[k][0] += float(row["Counter_Value"])
        if row["Dispatch_Id"] not in seen:
            seen.add(row["Dispatch_Id"])
            acc[k][1] += 1
            acc[k][2] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    return acc


def main():
    root = sys.argv[1]
    for w in ("cfg2", "vnet", "resunet", "unetr"):
        f, wr = per_kernel(os.path.join(root, "fetch_" + w), "FETCH_SIZE"), per_kernel(os.path.join(root, "write_" + w), "WRITE_SIZE")
        if not f:
            continue
        rows = []
        for k in f:
            n = max(1, f[k][1])
            fb, wb = 2.0 * f[k][0] * 1024.0 / n, (wr[k][0] * 1024.0 / max(1, wr[k][1])) if k in wr else 0.0
            ms = f[k][2] * 1e-6 / n
            rows.append((f[k][2], k, f[k][1], ms, fb, wb, (fb + wb) / (ms * 1e-3) / 1e12 if ms > 0 else 0.0))
        rows.sort(reverse=True)
        with open(os.path.join(root, f"traffic_{w}.csv"), "w") as fh:
            fh.write("kernel,launches,avg_ms,fetch_bytes_per_launch_corrected,write_bytes_per_launch,TB_per_s\n")
            for _, k, n, ms, fb, wb, tbs in rows:
                fh.write(f"\"{k}\",{n},{ms:.4f},{fb:.0f},{wb:.0f},{tbs:.2f}\n")
        print(w, "top:", [(r[1][:50], round(r[3], 3), round((r[4] + r[5]) / 1e6, 1)) for r in rows[:4]])


if __name__ == "__main__":
    main()
