#!/usr/bin/env python3
"""Lab driver: config 3 ande, one query (or --queries N) against 1 Mi rows, repeated; for rocprofv3 --kernel-trace."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1
unit = len(sys.argv) > 2 and sys.argv[2] == "unit"
q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, 2048, 40, unit), width=2048)
d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, 1 << 20, 2048, 40, unit), width=2048)
for _ in range(20):
    r = q.ande(d)
torch.cuda.synchronize()
print(r.total)
