#!/usr/bin/env python3
"""Lab driver: 4096 x 1 Mi rows @2048/40, one size op with a count width, five times; for rocprofv3 passes.
usage: run_narrow.py [ands|ors|xors|rms] [u8|u16|u32]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

op = sys.argv[1] if len(sys.argv) > 1 else "ands"
dt = {"u8": torch.uint8, "u16": torch.int16, "u32": None}[sys.argv[2] if len(sys.argv) > 2 else "u8"]
q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, 4096, 2048, 40, False), width=2048)
d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, 1 << 20, 2048, 40, False), width=2048)
out = None
for _ in range(5):
    out = getattr(q, op)(d, dtype=dt, out=out) if dt is not None else getattr(q, op)(d, out=out)
torch.cuda.synchronize()
print(int(out[:4, :1000].sum()))
