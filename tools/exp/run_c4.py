#!/usr/bin/env python3
"""Perf lab (not shipped): config 4 ande through a named kernel, three calls, for PMC runs.  usage: run_c4.py [algo]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

algo = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "auto" else None
w, act, nq, nd = 16384, 328, 64, 1 << 17
d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, nd, w, act, True), width=w)
q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, w, act, True), width=w)
for _ in range(3):
    r = q.ande(d, algo=algo)
torch.cuda.synchronize()
print("elements", r.total)
