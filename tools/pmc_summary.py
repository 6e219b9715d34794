#!/usr/bin/env python3
"""rocprofv3 --pmc passes (tools/pmc_mfma.sh) + a kernel-stats CSV -> per-kernel MFMA utilisation table (JSON + markdown) for profiles/.

    python tools/pmc_summary.py gpurun_out/pmc_mfma profiles/r02_a_kernel_stats.csv > profiles/r02_pmc_mfma.md

MFMA utilisation of a kernel = matrix-pipe busy cycles / (1024 SIMDs x kernel duration x clock), two ways:
  util_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * duration * clock)            (the counter is summed over the chip; = 32 x N for 32x32x16 MFMAs)
  util_issue  = 32 * SQ_INSTS_VALU_MFMA_F16 / (1024 * duration * clock)         (cross-check from the instruction count)
with clock = GRBM_GUI_ACTIVE / 8 / duration (guide: the counter is summed over the 8 XCDs; it reads high on dispatches shorter than ~0.3 ms).
SQ_BUSY_CYCLES is listed raw: its aggregation unit on gfx950 is not documented in the guide, so it is not used as a denominator.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(dirname):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name[:name.index("(")] if "(" in name else name


def main():
    pmc = load(sys.argv[1])
    dur = {}
    if len(sys.argv) > 2:
        for row in csv.DictReader(open(sys.argv[2])):
            dur[row["Name"]] = (float(row["AverageNs"]), int(row["Calls"]), float(row["Percentage"]))
    rows = []
    for k, c in pmc.items():
        if "SQ_BUSY_CYCLES" not in c or c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
            continue
        ns = dur.get(k, (None, None, None))[0]
        r = {"kernel": short(k), "avg_us": None if ns is None else round(ns / 1e3, 1), "pct_of_forward": dur.get(k, (0, 0, None))[2],
             "mfma_busy_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"], "sq_busy_cycles": c["SQ_BUSY_CYCLES"],
             "mfma_util_busy": None,
             "mfma_insts_f16": c.get("SQ_INSTS_VALU_MFMA_F16"), "valu_insts": c.get("SQ_INSTS_VALU"),
             "coexec_cycles": c.get("SQ_VALU_MFMA_COEXEC_CYCLES"), "wave_cycles": c.get("SQ_WAVE_CYCLES"), "waves": c.get("SQ_WAVES"),
             "wait_any": c.get("SQ_WAIT_ANY"), "wait_inst_any": c.get("SQ_WAIT_INST_ANY"), "active_inst_any": c.get("SQ_ACTIVE_INST_ANY"),
             "lds_bank_conflict": c.get("SQ_LDS_BANK_CONFLICT"), "lds_idx_active": c.get("SQ_LDS_IDX_ACTIVE"), "grbm_gui_active": c.get("GRBM_GUI_ACTIVE")}
        if ns and c.get("GRBM_GUI_ACTIVE") and c.get("SQ_INSTS_VALU_MFMA_F16"):
            clock_ghz = c["GRBM_GUI_ACTIVE"] / 8 / ns
            r["clock_ghz"] = round(clock_ghz, 3)
            r["mfma_util_issue"] = round(32 * c["SQ_INSTS_VALU_MFMA_F16"] / (1024 * ns * clock_ghz), 4)
            r["mfma_util_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * ns * clock_ghz), 4)
        rows.append(r)
    rows.sort(key=lambda r: -(r["pct_of_forward"] or 0))
    print("| kernel | avg us | % of forward | MFMA busy cycles / SIMD cycles | 32 x MFMA instructions / SIMD cycles | clock GHz | wait_any / wave_cycles | LDS conflict / active |")
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        wa = None if not (r["wait_any"] and r["wave_cycles"]) else round(r["wait_any"] / r["wave_cycles"], 3)
        lc = None if not (r["lds_idx_active"]) else round((r["lds_bank_conflict"] or 0) / r["lds_idx_active"], 3)
        print(f"| {r['kernel'][:70]} | {r['avg_us']} | {r['pct_of_forward']} | {r['mfma_util_busy']} | {r.get('mfma_util_issue')} | {r.get('clock_ghz')} | {wa} | {lc} |")
    json.dump(rows, open(os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), "pmc_mfma_summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
