#!/usr/bin/env python3
"""Perf lab (not shipped): config 4 ands only (timing experiments whose sizes may be wrong must not feed a fill)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

w, act, nq, nd = 16384, 328, 64, 1 << 17
d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, nd, w, act, True), width=w)
q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, w, act, True), width=w)
out = torch.empty((nq, nd), dtype=torch.int32, device="cuda")
best = 1e9
for _ in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    q.ands(d, out=out, algo="masktable")
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(f"ands masktable: {best:.3f} ms  checksum {int(out.sum(dtype=torch.int64))}", flush=True)
