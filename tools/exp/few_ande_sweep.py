#!/usr/bin/env python3
"""Lab: config 3 `ande` (1 Mi rows, 2048/40, EmptyData) for 1..32 queries -- where the position-mask path (few rows of A)
and the two-walk path meet.  Prints ms per call and ms per query."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

unit = len(sys.argv) > 1 and sys.argv[1] == "unit"
d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, 1 << 20, 2048, 40, False), width=2048)
for nq in (1, 2, 3, 4, 6, 8, 9, 12, 16, 24, 32):
    q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, 2048, 40, unit), width=2048)
    for _ in range(3):
        r = q.ande(d)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = q.ande(d)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"queries {nq:3d}  {best:.4f} ms  {best / nq * 1e3:.1f} us a query  elements {r.total}", flush=True)
