#!/usr/bin/env python3
"""Condenses gpurun_out/prof_ops_<tag>/ (written by tools/profile_ops.sh on the GPU box) into
profiles/<tag>_ops_profile.json: per case, every kernel of one element-op call with its average duration (kernel
trace) and its HBM bytes (FETCH_SIZE x 2 on gfx950 for wide reads, WRITE_SIZE; separate passes, KiB), the sums over
the call, and the algorithmic bytes of the call (inputs once + result CSR once)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALLS = 3   # tools/exp/run_one_op.py issues the op three times
SHAPES = {"c3": (2048, 40, False, 1 << 20), "c3u": (2048, 40, True, 1 << 20), "c4": (16384, 328, True, 1 << 17)}


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0]
    return name if len(name) <= 90 else name[:87] + "..."


def counter_means(case_dir, sub):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(case_dir, sub, "*", "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", f"prof_ops_{tag}")
    out = {"tag": tag, "method": "rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate "
                                 "passes (KiB); reads doubled (gfx950 tallies 16-B-per-lane reads at half); per element-op "
                                 "call = per-dispatch mean x dispatches per call", "cases": {}}
    for case_dir in sorted(glob.glob(os.path.join(src, "*_q*_*"))):
        case = os.path.basename(case_dir)
        shape, q, op = case.split("_")
        nq = int(q[1:])
        width, active, unit, nd = SHAPES[shape]
        elements = None
        for line in open(os.path.join(case_dir, "kt.log")):
            if line.startswith("elements"):
                elements = int(line.split()[1])
        eb = 8 if unit else 4
        algorithmic = (nd + nq) * (eb * active + 4) + (nq * nd + 1) * 4 + (elements or 0) * eb
        stats = glob.glob(os.path.join(case_dir, "kt", "*", "*_kernel_stats.csv"))
        kernels = {}
        # the kernels of the set-up (the two uploads' checks, the runtime's copies and fills) are not part of a call
        setup = ("k_check_flat", "k_check_rows", "k_fold_findings", "k_neutral_findings", "k_locate_findings", "__amd_rocclr")
        for r in csv.DictReader(open(stats[0])) if stats else []:
            if any(x in r["Name"] for x in setup):
                continue
            kernels[short(r["Name"])] = {"dispatches_per_call": int(r["Calls"]) / CALLS, "avg_ns": float(r["AverageNs"])}
        pmc = {}
        for sub in glob.glob(os.path.join(case_dir, "pmc_*")):
            if os.path.isdir(sub):
                for k, cs in counter_means(case_dir, os.path.basename(sub)).items():
                    pmc.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
        total_ns = read_b = write_b = 0.0
        for k, info in kernels.items():
            c = pmc.get(k, {})
            n = info["dispatches_per_call"]
            info["ns_per_call"] = info["avg_ns"] * n
            if "FETCH_SIZE" in c:
                info["hbm_read_bytes_per_call"] = c["FETCH_SIZE"] * 1024 * 2 * n
                read_b += info["hbm_read_bytes_per_call"]
            if "WRITE_SIZE" in c:
                info["hbm_write_bytes_per_call"] = c["WRITE_SIZE"] * 1024 * n
                write_b += info["hbm_write_bytes_per_call"]
            sq = {x: c[x] for x in c if x.startswith("SQ_")}
            if sq:
                info["sq_per_dispatch"] = sq
            total_ns += info["ns_per_call"]
        out["cases"][case] = {
            "queries": nq, "rows": nd, "width": width, "active": active, "unit_payloads": unit, "elements": elements,
            "algorithmic_bytes": algorithmic, "kernel_ns_per_call": total_ns,
            "hbm_read_bytes_per_call": read_b, "hbm_write_bytes_per_call": write_b,
            "traffic_over_algorithmic": (read_b + write_b) / algorithmic if algorithmic else None,
            "achieved_GBps_kernel_time": algorithmic / total_ns if total_ns else None,
            "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["ns_per_call"])),
        }
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    path = os.path.join(ROOT, "profiles", f"{tag}_ops_profile.json")
    json.dump(out, open(path, "w"), indent=1)
    for case, c in out["cases"].items():
        print(case, "kernel us/call %.1f" % (c["kernel_ns_per_call"] / 1e3), "alg MB %.1f" % (c["algorithmic_bytes"] / 1e6),
              "read MB %.1f write MB %.1f" % (c["hbm_read_bytes_per_call"] / 1e6, c["hbm_write_bytes_per_call"] / 1e6))
        for k, i in list(c["kernels"].items())[:6]:
            print("   %-70s x%.0f %8.1f us" % (k[:70], i["dispatches_per_call"], i["ns_per_call"] / 1e3))


if __name__ == "__main__":
    main()
