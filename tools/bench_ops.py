#!/usr/bin/env python3
"""Secondary measurements (not the headline metric): element ops over 1 Mi rows (BASELINE.json
configs[2]) and the UnitData 16384/328 shape (configs[3]), per-op time and achieved bytes/s
against algorithmic bytes (inputs once + output CSR once).  Prints one JSON line per case."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(reps):
        ev0.record()
        r = fn()
        ev1.record()
        torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1))
    return best, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1 << 20)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cases", default="c3,c4")
    args = ap.parse_args()
    cases = []
    if "c3" in args.cases:
        for nq in (1, 16):
            cases.append(("c3", nq, args.rows, 2048, 40, False))
    if "c4" in args.cases:
        cases.append(("c4", 64, args.rows // 8, 16384, 328, True))
    for tag, nq, nd, width, active, unit in cases:
        q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, width, active, unit), width=width)
        d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, nd, width, active, unit), width=width)
        eb = 8 if unit else 4
        in_bytes = nd * (eb * active + 4) + nq * (eb * active + 4)
        for sname, ename in (("ands", "ande"), ("ors", "ore"), ("xors", "xore"), ("rms", "rme")):
            ms_s, counts = timed(lambda: getattr(q, sname)(d), args.reps)
            ms_e, res = timed(lambda: getattr(q, ename)(d), args.reps)
            out_bytes = nq * nd * 4 + res.total * eb
            print(json.dumps({
                "case": tag, "queries": nq, "rows": nd, "width": width, "active": active, "unit": unit,
                "size_op": sname, "size_ms": round(ms_s, 4),
                "size_GBps": round((in_bytes + nq * nd * 4) / ms_s / 1e6, 1),
                "elem_op": ename, "elem_ms": round(ms_e, 4), "elements": res.total,
                "elem_GBps": round((in_bytes + out_bytes) / ms_e / 1e6, 1),
                "pair_ops_per_s": round(nq * nd / (ms_e * 1e-3), 0)}), flush=True)
            del res, counts


if __name__ == "__main__":
    main()
