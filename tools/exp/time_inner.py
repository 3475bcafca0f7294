#!/usr/bin/env python3
"""Perf lab (not shipped): times sdrb_inner on the 2048/40 shape.  usage: time_inner.py [queries] [rows] [algo] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
algo = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "auto" else None
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
width, active = 2048, 40
rng = np.random.default_rng(0)
d_off, d_ids, _ = sdr_amd.synth_csr(1, 0, nd, width, active)
d = SDRBatch.from_csr(d_off, d_ids, rng.standard_normal(d_ids.size).astype(np.float32), width=width, kind="arith")
q_off, q_ids, _ = sdr_amd.synth_csr(2, 0, nq, width, active)
q = SDRBatch.from_csr(q_off, q_ids, rng.standard_normal(q_ids.size).astype(np.float32), width=width, kind="arith")
out = torch.empty((nq, nd), dtype=torch.float32, device="cuda:0")
q.inner(d, out=out, algo=algo)
torch.cuda.synchronize()
best = 1e9
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    q.inner(d, out=out, algo=algo)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
nbytes = nd * (8 * active + 4) + nq * (8 * active + 4) + nq * nd * 4
print(f"inner {nq} x {nd} algo={algo}: {best:.3f} ms  {nbytes / best / 1e6:.0f} GB/s", flush=True)
