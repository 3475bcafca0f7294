#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag>/ directory (written by tools/profile_bench.sh on the GPU
box) into profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json and, for the headline
kernel, profiles/hbm_traffic.json (read by bench.py for roofline.traffic).

HBM bytes follow the MI355X guide's recipe: FETCH_SIZE and WRITE_SIZE are collected in
separate passes and reported in KiB; on gfx950 FETCH_SIZE counts a wide (16 B per lane)
coalesced read at half its bytes, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane
stores and is calibrated here for this kernel's dword-per-lane store shape on a pattern of
known size (tools/exp/lab_store cal)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "k_size_bitslice"


def counters(path_glob, kernel_substr):
    """{counter: [value per dispatch]} for dispatches whose kernel name contains the substring; when the run holds
    launches of several sizes (the bench also times smaller configurations) only those with the largest grid -- the
    headline shape -- are kept."""
    rows = []
    for f in glob.glob(path_glob):
        for row in csv.DictReader(open(f)):
            if kernel_substr in row["Kernel_Name"]:
                rows.append(row)
    biggest = max((int(r["Grid_Size"]) for r in rows), default=0)
    out = defaultdict(list)
    for row in rows:
        if int(row["Grid_Size"]) == biggest:
            out[row["Counter_Name"]].append(float(row["Counter_Value"]))
    return out


def mean(xs):
    return sum(xs) / len(xs) if xs else None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)

    # kernel stats (rocprofv3 --kernel-trace --stats)
    stats = glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            name = r["Name"]
            name = name if len(name) < 160 else name[:157] + "..."
            w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    main_row = next(r for r in rows if KERNEL in r["Name"])
    avg_ns = float(main_row["AverageNs"])
    # the stats row averages every launch of the kernel; the trace tells the headline-shaped launches apart
    trace = glob.glob(os.path.join(src, "kt", "*", "*_kernel_trace.csv"))
    if trace:
        t_rows = [r for r in csv.DictReader(open(trace[0])) if KERNEL in r["Kernel_Name"]]
        if t_rows:
            size = lambda r: int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])   # noqa: E731
            biggest = max(size(r) for r in t_rows)
            durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in t_rows if size(r) == biggest]
            avg_ns = sum(durs) / len(durs)
            main_row = dict(main_row, Calls=str(len(durs)))

    pmc = {}
    for d in glob.glob(os.path.join(src, "pmc_*")):
        if os.path.isdir(d):
            for k, v in counters(os.path.join(d, "*", "*_counter_collection.csv"), KERNEL).items():
                pmc[k] = mean(v)
    cal = counters(os.path.join(src, "cal_WRITE_SIZE", "*", "*_counter_collection.csv"), "k_tiles")
    cal_bytes = 4096 * (1 << 20) * 4
    cal_factor = cal_bytes / (mean(cal["WRITE_SIZE"]) * 1024) if cal.get("WRITE_SIZE") else None

    fetch_bytes = pmc["FETCH_SIZE"] * 1024 * 2          # gfx950: wide reads tallied at half
    write_raw = pmc["WRITE_SIZE"] * 1024
    write_bytes = write_raw * (cal_factor if cal_factor else 1.0)
    summary = {
        "tag": tag, "kernel": main_row["Name"][:120], "calls": int(main_row["Calls"]),
        "avg_kernel_ns": avg_ns,
        "pmc_mean_per_launch": pmc,
        "write_size_calibration": {"pattern_bytes": cal_bytes,
                                   "WRITE_SIZE_KiB": mean(cal["WRITE_SIZE"]) if cal.get("WRITE_SIZE") else None,
                                   "factor": cal_factor},
        "hbm_read_bytes_per_launch": fetch_bytes,
        "hbm_write_bytes_per_launch": write_bytes,
        "hbm_write_bytes_uncalibrated": write_raw,
    }
    if "GRBM_GUI_ACTIVE" in pmc:
        summary["effective_clock_GHz"] = pmc["GRBM_GUI_ACTIVE"] / 8 / avg_ns
    if "TCC_HIT_sum" in pmc:
        summary["l2_hit_rate"] = pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])
    json.dump(summary, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
    json.dump({"queries": 4096, "rows": 1 << 20, "op": "ands", "tag": tag,
               "hbm_bytes_per_launch": fetch_bytes + write_bytes,
               "read": fetch_bytes, "write": write_bytes,
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB; FETCH_SIZE x2 "
                         "(gfx950 wide-read correction); WRITE_SIZE x calibration factor measured on a "
                         "17.18 GB dword-per-lane store pattern"},
              open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
