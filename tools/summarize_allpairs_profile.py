#!/usr/bin/env python3
"""Condenses gpurun_out/prof_ap_<tag>/ (tools/profile_allpairs.sh) into
profiles/<tag>_allpairs_kernel_stats.csv and profiles/<tag>_allpairs_pmc.json."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "k_allpairs_mfma"      # overridden by a second argument (e.g. k_allpairs_bitslice)


def main():
    global KERNEL
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    if len(sys.argv) > 2:
        KERNEL = sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", f"prof_ap_{tag}")
    dst = os.path.join(ROOT, "profiles")
    stats = glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(dst, f"{tag}_allpairs_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"][:150], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    avg_ns = float(next(r for r in rows if KERNEL in r["Name"])["AverageNs"])
    pmc = defaultdict(list)
    for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if KERNEL in row["Kernel_Name"]:
                pmc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    # the first dispatch of a process also builds the byte matrices; the kernel itself is identical, keep the mean
    out = {k: sum(v) / len(v) for k, v in pmc.items()}
    out["kernel_avg_ns"] = avg_ns
    if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "SQ_BUSY_CYCLES" in out:
        out["note"] = "SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* quad-cycles"
    json.dump(out, open(os.path.join(dst, f"{tag}_allpairs_pmc.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
