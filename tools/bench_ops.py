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
    ap.add_argument("--only", default="", help="comma-separated size-op names to run (default: all four)")
    ap.add_argument("--algo", default=None, help="force a kernel family for the element ops (e.g. merge, rank, rowtable)")
    args = ap.parse_args()
    cases = []
    if "c3" in args.cases.split(","):
        for nq in (1, 16):
            cases.append(("c3", nq, args.rows, 2048, 40, False))
    if "c3u" in args.cases:      # the configs[2] shape with UnitData payloads on both sides
        for nq in (1, 16):
            cases.append(("c3u", nq, args.rows, 2048, 40, True))
    if "c4" in args.cases.split(","):
        cases.append(("c4", 64, args.rows // 8, 16384, 328, True))
    if "c4q" in args.cases:      # configs[3] with 24 queries, so that the unions fit 32-bit offsets
        cases.append(("c4q", 24, args.rows // 8, 16384, 328, True))
    only = set(args.only.split(",")) if args.only else None
    for tag, nq, nd, width, active, unit in cases:
        q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, width, active, unit), width=width)
        d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, nd, width, active, unit), width=width)
        eb = 8 if unit else 4
        in_bytes = nd * (eb * active + 4) + nq * (eb * active + 4)
        # ors never looks at payloads (SDR.hpp:1298-1301): its kernel reads the ids only
        in_bytes_ids = nd * (4 * active + 4) + nq * (4 * active + 4)
        for sname, ename in (("ands", "ande"), ("ors", "ore"), ("xors", "xore"), ("rms", "rme")):
            if only and sname not in only:
                continue
            ms_s, counts = timed(lambda: getattr(q, sname)(d), args.reps)
            try:
                ms_e, res = timed(lambda: getattr(q, ename)(d, algo=args.algo), args.reps)
            except sdr_amd.SdrbError as e:      # e.g. a union too large for 32-bit offsets: size op only
                print(json.dumps({"case": tag, "queries": nq, "rows": nd, "size_op": sname, "size_ms": round(ms_s, 4),
                                  "size_GBps": round(((in_bytes_ids if sname == "ors" else in_bytes) + nq * nd * 4) / ms_s / 1e6, 1),
                                  "elem_op": ename, "skipped": str(e)[:80]}), flush=True)
                continue
            out_bytes = (nq * nd + 1) * 4 + res.total * eb
            print(json.dumps({
                "case": tag, "queries": nq, "rows": nd, "width": width, "active": active, "unit": unit,
                "size_op": sname, "size_ms": round(ms_s, 4),
                "size_GBps": round(((in_bytes_ids if sname == "ors" else in_bytes) + nq * nd * 4) / ms_s / 1e6, 1),
                "elem_op": ename, "elem_ms": round(ms_e, 4), "elements": res.total,
                "elem_GBps": round((in_bytes + out_bytes) / ms_e / 1e6, 1),
                "pair_ops_per_s": round(nq * nd / (ms_e * 1e-3), 0)}), flush=True)
            del res, counts
    if "widthless" in args.cases:
        # The same shapes with NO declared width (the reference's ids are arbitrary ints): the per-pair merge kernel
        # against the default choice (small labels, then the table kernels), and the declared-width time beside them.
        shapes = [("c2 slab", 64, args.rows, 2048, 40, False, "ands"), ("c3", 16, args.rows, 2048, 40, False, "ands"),
                  ("c3 one query", 1, args.rows, 2048, 40, False, "ands"), ("c2 512 queries", 512, args.rows, 2048, 40, False, "ands"),
                  ("c4", 64, args.rows // 8, 16384, 328, True, "ands"), ("c4 ors", 64, args.rows // 8, 16384, 328, True, "ors"),
                  ("unit 2048/40", 64, args.rows, 2048, 40, True, "ands")]
        for tag, nq, nd, width, active, unit, sname in shapes:
            q_csr = sdr_amd.synth_csr(2, 0, nq, width, active, unit)
            d_csr = sdr_amd.synth_csr(1, 0, nd, width, active, unit)
            out = torch.empty((nq, nd), dtype=torch.int32, device="cuda:0")
            times = {}
            # "undeclared": no width given, no negative id -- the library derives one (largest id + 1);
            # "moved": the same rows with every id moved down by 1000 -- no width to derive, but ids close together, which
            #          the library moves to start at zero for the call;
            # "widthless": ids far apart and partly negative (times 100003, less 10^9): small labels, then the table kernels
            far = lambda x: (x.astype(np.int64) * 100003 - 1000000000).astype(np.int32)   # noqa: E731
            ename = "ande" if sname == "ands" else "ore"
            for label, w, f in (("declared", width, lambda x: x), ("undeclared", 0, lambda x: x), ("moved", 0, lambda x: x - 1000),
                                ("widthless", 0, far)):
                q = SDRBatch.from_csr(q_csr[0], f(q_csr[1]), q_csr[2], width=w)
                d = SDRBatch.from_csr(d_csr[0], f(d_csr[1]), d_csr[2], width=w)
                times[label], _ = timed(lambda: getattr(q, sname)(d, out=out), args.reps)
                if label == "widthless":
                    ref = out.clone()
                    times["merge"], _ = timed(lambda: getattr(q, sname)(d, out=out, algo="merge"), max(1, args.reps // 2))
                    same = bool(torch.equal(ref, out))
                if label in ("declared", "moved", "widthless") and nq <= 64 and ename == "ande":
                    # the elements too: ids moved to zero take the table kernels, ids as they are the per-pair merge
                    times[label + "_elems"], res = timed(lambda: getattr(q, ename)(d), max(1, args.reps // 2))
                    if label == "moved":
                        times["merge_elems"], res_m = timed(lambda: getattr(q, ename)(d, algo="merge"), max(1, args.reps // 4))
                        elems_equal = bool(torch.equal(res.offsets, res_m.offsets) and torch.equal(res.ids, res_m.ids))
                        del res_m
                    if label == "widthless":    # ids far apart: a wave per pair with binary searches against the lane-per-pair merge
                        times["merge_far_elems"], res_m = timed(lambda: getattr(q, ename)(d, algo="merge"), max(1, args.reps // 4))
                        far_equal = bool(torch.equal(res.offsets, res_m.offsets) and torch.equal(res.ids, res_m.ids))
                        del res_m
                    del res
                del q, d
            eb = 8 if unit and sname != "ors" else 4
            nbytes = (nd + nq) * (eb * active + 4) + nq * nd * 4
            line = {"case": "widthless", "shape": tag, "queries": nq, "rows": nd, "width": width, "active": active,
                    "unit": unit, "op": sname, "declared_ms": round(times["declared"], 4),
                    "undeclared_ms": round(times["undeclared"], 4), "moved_ms": round(times["moved"], 4),
                    "widthless_ms": round(times["widthless"], 4), "merge_ms": round(times["merge"], 4),
                    "speedup_over_merge": round(times["merge"] / times["widthless"], 1), "equal": same,
                    "widthless_GBps": round(nbytes / times["widthless"] / 1e6, 1),
                    "merge_pair_ops_per_s": round(nq * nd / (times["merge"] * 1e-3), 0)}
            if "moved_elems" in times:
                line.update({"elem_op": ename, "declared_elems_ms": round(times["declared_elems"], 4),
                             "moved_elems_ms": round(times["moved_elems"], 4), "merge_elems_ms": round(times["merge_elems"], 4),
                             "elems_equal": elems_equal, "widthless_elems_ms": round(times["widthless_elems"], 4),
                             "merge_far_elems_ms": round(times["merge_far_elems"], 4), "far_elems_equal": far_equal})
            print(json.dumps(line), flush=True)
    if "inner" in args.cases:
        # sparse dot products (SDR::inner / diff_mul): 4096 x rows pairs at 2048/40 with float payloads,
        # and the matrix x vector shape (1 x rows)
        width, active = 2048, 40
        d_off, d_ids, _ = sdr_amd.synth_csr(1, 0, args.rows, width, active)
        rng = np.random.default_rng(0)
        d = SDRBatch.from_csr(d_off, d_ids, rng.standard_normal(d_ids.size).astype(np.float32), width=width, kind="arith")
        for nq in (1, 4096):
            q_off, q_ids, _ = sdr_amd.synth_csr(2, 0, nq, width, active)
            q = SDRBatch.from_csr(q_off, q_ids, rng.standard_normal(q_ids.size).astype(np.float32), width=width, kind="arith")
            out = torch.empty((nq, args.rows), dtype=torch.float32, device="cuda:0")
            ms, _ = timed(lambda: q.inner(d, out=out), args.reps)
            nbytes = args.rows * (8 * active + 4) + nq * (8 * active + 4) + nq * args.rows * 4
            print(json.dumps({"case": "inner", "queries": nq, "rows": args.rows, "width": width, "active": active,
                              "ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 1),
                              "pair_ops_per_s": round(nq * args.rows / (ms * 1e-3), 0)}), flush=True)


if __name__ == "__main__":
    main()
