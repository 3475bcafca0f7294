#!/usr/bin/env python3
"""Lab: the bit-sliced kernel for long rows against the kernels it replaces (mask table, popcount), id-only ops."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


shapes = [(64, 1 << 17, 16384, 328), (4096, 1 << 17, 16384, 328), (64, 1 << 20, 2048, 100), (1024, 1 << 18, 2048, 300),
          (64, 1 << 18, 8192, 160), (512, 1 << 18, 4096, 64), (64, 1 << 16, 32768, 650)]
for nq, nd, w, act in shapes:
    q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, w, act), width=w)
    d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, nd, w, act), width=w)
    out = torch.empty((nq, nd), dtype=torch.int32, device="cuda")
    nbytes = (nd + nq) * (4 * act + 4) + nq * nd * 4
    row = {"queries": nq, "rows": nd, "width": w, "active": act, "floor_us_at_8TBps": round(nbytes / 8e12 * 1e6, 1)}
    for op in ("ors", "ands"):
        for algo in ("bitslice_long", "masktable", "popcount", "bitslice"):
            for mb in (("4", "8") if algo == "bitslice_long" else ("",)):
                if mb:
                    os.environ["SDRB_LONG_MASK_BYTES"] = mb
                try:
                    ms = timed(lambda: getattr(q, op)(d, out=out, algo=algo))
                    row[f"{op}_{algo}{mb}_ms"] = round(ms, 4)
                    row[f"{op}_{algo}{mb}_frac"] = round(nbytes / (ms * 1e-3) / 8e12, 3)
                except sdr_amd.SdrbError:
                    pass
    os.environ.pop("SDRB_LONG_MASK_BYTES", None)
    print(json.dumps(row), flush=True)
