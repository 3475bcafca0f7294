#!/usr/bin/env python3
"""Perf lab (not shipped): payload-dependent size ops of a few UnitData queries x 1 Mi rows: default kernel choice
against the lane-per-row size kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

nd = 1 << 20
d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, nd, 2048, 40, True), width=2048)
for nq in (1, 2, 4, 8, 16, 17, 32):
    q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, 2048, 40, True), width=2048)
    out = torch.empty((nq, nd), dtype=torch.int32, device="cuda:0")
    line = f"Q={nq:3d}"
    for name in ("ands", "xors"):
        for algo in (None, "lanerows"):
            fn = lambda: getattr(q, name)(d, out=out, algo=algo)   # noqa: E731
            fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            line += f"  {name}/{algo}={best:.4f}"
    print(line, flush=True)
