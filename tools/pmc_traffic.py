#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE passes (tools/prof_r03.sh) + a kernel-stats CSV -> per-kernel HBM traffic and rate table (markdown).

    python tools/pmc_traffic.py gpurun_out/r03/cfg2_pmc gpurun_out/r03/cfg2_kernel_stats.csv

Per the guide (MI355X_MICROARCH.md, HBM): on gfx950 FETCH_SIZE (KB) reports exactly half of the bytes of a wide coalesced streaming read -
both readings are listed ("fetch x1" = as reported, "fetch x2" = corrected); WRITE_SIZE reads streaming stores exactly.  Rate = bytes per
launch / average launch duration from the kernel-stats run of the same command (a different run: the counter passes serialise nothing, but
durations under --pmc are not used).  The "upper" rate uses fetch x2 + write, the "lower" fetch x1 + write.
"""
import csv, glob, os, sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name[:name.index("(")] if "(" in name else name


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    dur = {row["Name"]: (float(row["AverageNs"]), int(row["Calls"]), float(row["Percentage"])) for row in csv.DictReader(open(sys.argv[2]))}
    rows = []
    for k, c in acc.items():
        if k not in dur or not c.get("FETCH_SIZE"):
            continue
        f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024.0
        w = sum(c.get("WRITE_SIZE", [0])) / max(1, len(c.get("WRITE_SIZE", [0]))) * 1024.0
        ns = dur[k][0]
        rows.append((dur[k][2], short(k), ns / 1e3, f / 1e6, 2 * f / 1e6, w / 1e6, (f + w) / ns / 1e3, (2 * f + w) / ns / 1e3))
    rows.sort(reverse=True)
    print("| kernel | avg us | % of run | fetch x1 MB | fetch x2 MB | write MB | lower TB/s | upper TB/s |")
    print("|---|---|---|---|---|---|---|---|")
    for pct, k, us, f1, f2, w, lo, up in rows:
        if pct < 0.2:
            continue
        print(f"| {k[:64]} | {us:.1f} | {pct:.2f} | {f1:.0f} | {f2:.0f} | {w:.0f} | {lo:.2f} | {up:.2f} |")


if __name__ == "__main__":
    main()
