#!/usr/bin/env python3
"""Perf lab (not shipped): config 4 (64 x 128 Ki rows, 16384/328, UnitData) ands / ande through named kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

w, act, nq, nd = 16384, 328, 64, 1 << 17
d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, nd, w, act, True), width=w)
q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, w, act, True), width=w)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best, r


for algo in (None, "masktable"):
    ms, _ = timed(lambda: q.ands(d, algo=algo))
    print(f"ands algo={algo}: {ms:.3f} ms", flush=True)
for algo in (None,):
    ms, r = timed(lambda: q.ande(d, algo=algo))
    print(f"ande algo={algo}: {ms:.3f} ms  elements {r.total}", flush=True)
