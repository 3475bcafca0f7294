#!/usr/bin/env python3
"""Lab: one `ande` of ids far apart (64 x 1 Mi rows, 2048/40) through the wave-per-pair search fill, for counter passes."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

nq, nd = 64, 1 << 20
far = lambda x: (x.astype(np.int64) * 100003 - 1000000000).astype(np.int32)   # noqa: E731
q_csr = sdr_amd.synth_csr(2, 0, nq, 2048, 40)
d_csr = sdr_amd.synth_csr(1, 0, nd, 2048, 40)
q = SDRBatch.from_csr(q_csr[0], far(q_csr[1]), None)
d = SDRBatch.from_csr(d_csr[0], far(d_csr[1]), None)
for _ in range(3):
    r = q.ande(d, algo="search")
    print(r.total)
torch.cuda.synchronize()
