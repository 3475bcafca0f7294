#!/usr/bin/env python3
"""Perf lab (not shipped): one element op on one shape, a few launches, for PMC runs.
usage: run_one_op.py <case> <queries> <op> [algo]   case: c3 | c3u | c4"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402,F401

import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

case, nq, op = sys.argv[1], int(sys.argv[2]), sys.argv[3]
algo = sys.argv[4] if len(sys.argv) > 4 else None
width, active, unit, nd = {"c3": (2048, 40, False, 1 << 20), "c3u": (2048, 40, True, 1 << 20),
                           "c4": (16384, 328, True, 1 << 17)}[case]
q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, width, active, unit), width=width)
d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, nd, width, active, unit), width=width)
for _ in range(3):
    r = getattr(q, op)(d, algo=algo)
torch.cuda.synchronize()
print("elements", r.total)
