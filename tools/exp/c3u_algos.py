#!/usr/bin/env python3
"""Lab: UnitData element ops of a few queries against 1 Mi rows (2048/40) through each fill kernel that takes them."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch, SdrbError  # noqa: E402

d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, 1 << 20, 2048, 40, True), width=2048)
for nq in (1, 2, 3, 4):
    q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, 2048, 40, True), width=2048)
    for op in ("ore", "rme", "xore"):
        line = f"q={nq} {op}:"
        for algo in (None, "rank", "lanemerge", "merge", "bitmap"):
            try:
                getattr(q, op)(d, algo=algo)
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    getattr(q, op)(d, algo=algo)
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1))
                line += f"  {algo}={best:.3f}"
            except SdrbError:
                line += f"  {algo}=n/a"
        print(line, flush=True)
