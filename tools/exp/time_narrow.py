#!/usr/bin/env python3
"""Lab: 4096 x 1 Mi rows @2048/40: the size ops with byte, 16-bit and 32-bit counts, ms per call (events, best of 7)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, 4096, 2048, 40, False), width=2048)
d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, 1 << 20, 2048, 40, False), width=2048)
for name, dt in (("u8", torch.uint8), ("u16", torch.int16), ("u32", None)):
    for op in ("ands", "ors", "xors"):
        if name == "u8" and op == "ors":
            continue   # 80 fits a byte, kept out only to keep the list short
        out = None
        best = 1e9
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = getattr(q, op)(d, dtype=dt, out=out) if dt is not None else getattr(q, op)(d, out=out)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(f"{op}_{name}: {best:.4f} ms", flush=True)
        del out
        torch.cuda.empty_cache()
