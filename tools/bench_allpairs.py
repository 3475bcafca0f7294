#!/usr/bin/env python3
"""All-pairs overlap + per-row top-k: one GPU's share of the 256 Ki x 256 Ki problem
(rows [0, na) of the database against all nb rows), width 2048 / 40 active.

Prints one JSON line: rows/s, pair-ops/s, int8 MFMA utilisation (2 * na * nb * width ops over
the measured time against the dense int8 peak) and a parity spot check against the oracle.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

I8_PEAK_OPS = 5.03e15   # 256 CUs x 4 SIMDs x 2048 int8 ops/clk x 2.4 GHz (2x the bf16 dense rate)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--na", type=int, default=32768)
    ap.add_argument("--nb", type=int, default=262144)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--active", type=int, default=40)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--check-rows", type=int, default=4)
    ap.add_argument("--algo", default=None, choices=[None, "mfma", "bitslice"])
    args = ap.parse_args()

    import torch
    import sdr_amd
    from sdr_amd import SDRBatch

    off, ids, _ = sdr_amd.synth_csr(1, 0, args.nb, args.width, args.active)
    b = SDRBatch.from_csr(off, ids, None, width=args.width)
    a_off = off[: args.na + 1].copy()
    a = SDRBatch.from_csr(a_off, ids[: int(a_off[-1])], None, width=args.width)
    a.allpairs_topk(b, args.k, exclude_offset=0, algo=args.algo)      # builds the byte matrices, warms up
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        counts, rows = a.allpairs_topk(b, args.k, exclude_offset=0, algo=args.algo)
    ev1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.steps
    dev = ev0.elapsed_time(ev1) * 1e-3 / args.steps

    ok = None
    if args.check_rows:
        import oracle
        pick = np.linspace(0, args.na - 1, args.check_rows).astype(np.int64)
        q_off = np.concatenate([[0], np.cumsum(off[pick + 1] - off[pick])]).astype(np.uint64)
        q_ids = np.concatenate([ids[int(off[r]):int(off[r + 1])] for r in pick])
        want = oracle.port().size_op(oracle.AND, oracle.Batch(q_off, q_ids, None), oracle.Batch(off, ids, None))
        ok = True
        gc, gr = counts.cpu().numpy(), rows.cpu().numpy()
        for n, i in enumerate(pick):
            c = want[n].astype(np.int64)
            cand = np.arange(args.nb)
            cand = cand[cand != i]
            order = cand[np.lexsort((cand, -c[cand]))][: args.k]
            ok = ok and np.array_equal(gr[i], order) and np.array_equal(gc[i], c[order])
    ops = 2.0 * args.na * args.nb * args.width
    print(json.dumps({
        "workload": f"all-pairs overlap + top-{args.k}: {args.na} x {args.nb} rows, width {args.width}, "
                    f"{args.active} active, self match excluded",
        "algo": args.algo or "default", "ms": dev * 1e3, "wall_ms": wall * 1e3,
        "pair_ops_per_s": args.na * args.nb / dev,
        "int8_ops_per_s": ops / dev, "mfma_peak_ops_per_s": I8_PEAK_OPS, "mfma_frac": ops / dev / I8_PEAK_OPS,
        "parity_checked": bool(ok) if ok is not None else None,
    }))
    if ok is False:
        raise SystemExit("parity check failed")


if __name__ == "__main__":
    main()
