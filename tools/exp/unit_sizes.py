#!/usr/bin/env python3
"""Perf lab (not shipped): size ops with UnitData payloads at the headline shape (Q x 1 Mi rows, 2048/40)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nd = 1 << 20
q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, 2048, 40, True), width=2048)
d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, nd, 2048, 40, True), width=2048)
out = torch.empty((nq, nd), dtype=torch.int32, device="cuda:0")
for name in ("ands", "ors", "xors", "rms"):
    for algo in (None, "masktable"):
        fn = lambda: getattr(q, name)(d, out=out, algo=algo)   # noqa: E731
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        print(name, algo, nq, f"{e0.elapsed_time(e1):.3f} ms", flush=True)
