#!/usr/bin/env python3
"""Perf lab (not shipped): where the lane-per-row match kernels win -- UnitData ands (lanerows vs masktable) and dot
products (sparse vs table) at several widths / row lengths, 1024 queries x 256 Ki rows."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

nq, nd = 1024, 1 << 18


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


rng = np.random.default_rng(0)
out = torch.empty((nq, nd), dtype=torch.int32, device="cuda:0")
for width, active in ((2048, 40), (2048, 64), (1024, 40), (4096, 64), (512, 40), (4096, 20), (256, 40)):
    d_csr = sdr_amd.synth_csr(1, 0, nd, width, active, True)
    q_csr = sdr_amd.synth_csr(2, 0, nq, width, active, True)
    d, q = SDRBatch.from_csr(*d_csr, width=width), SDRBatch.from_csr(*q_csr, width=width)
    line = f"{width}/{active} mean bits per id of a 32-row tile {32 * active / width:.2f}: ands"
    for algo in ("lanerows", "masktable", None):
        line += f"  {algo}={timed(lambda: q.ands(d, out=out, algo=algo)):.3f}"
    da = SDRBatch.from_csr(d_csr[0], d_csr[1], rng.standard_normal(d_csr[1].size).astype(np.float32), width=width, kind="arith")
    qa = SDRBatch.from_csr(q_csr[0], q_csr[1], rng.standard_normal(q_csr[1].size).astype(np.float32), width=width, kind="arith")
    of = out.view(torch.float32)
    line += " | inner"
    for algo in ("sparse", "table", None):
        line += f"  {algo}={timed(lambda: qa.inner(da, out=of, algo=algo)):.3f}"
    print(line, flush=True)
