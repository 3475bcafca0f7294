#!/usr/bin/env python3
"""Lab (not shipped): time of the offsets scan inside a 16-query ande over 1 Mi rows (kernel trace not needed:
the whole op is timed, variants differ only in the scan)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, 16, 2048, 40), width=2048)
d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, 1 << 20, 2048, 40), width=2048)
for _ in range(3):
    q.ande(d)
torch.cuda.synchronize()
best = 1e9
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    q.ande(d)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(f"ande 16 x 1 Mi: {best * 1e3:.1f} us")
