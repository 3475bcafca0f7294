#!/usr/bin/env python3
"""Perf lab (not shipped): one payload-dependent size op at Q x 1 Mi rows, 2048/40, for counter passes.
usage: run_unit_size.py [Q] [op] [algo] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
op = sys.argv[2] if len(sys.argv) > 2 else "ands"
algo = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "default" else None
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
nd = 1 << 20
q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, 2048, 40, True), width=2048)
d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, nd, 2048, 40, True), width=2048)
out = torch.empty((nq, nd), dtype=torch.int32, device="cuda:0")
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    getattr(q, op)(d, out=out, algo=algo)
    e1.record()
    torch.cuda.synchronize()
    print(op, algo, nq, f"{e0.elapsed_time(e1):.3f} ms", flush=True)
