#!/usr/bin/env python3
"""Perf lab (not shipped): one query against 1 Mi rows (2048/40), every size kernel that takes the shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1
unit = len(sys.argv) > 2 and sys.argv[2] == "unit"
nd = 1 << 20
q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, 2048, 40, unit), width=2048)
d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, nd, 2048, 40, unit), width=2048)
out = torch.empty((nq, nd), dtype=torch.int32, device="cuda:0")
bytes_alg = nd * 164 + nq * 164 + nq * nd * 4
for name in ("ands", "xors"):
    for algo in (None, "bitslice", "popcount", "rowtable", "merge"):
        try:
            fn = lambda: getattr(q, name)(d, out=out, algo=algo)   # noqa: E731
            fn()
        except sdr_amd.SdrbError as e:
            print(name, algo, "refused", e.code)
            continue
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(name, algo, nq, f"{best * 1e3:.1f} us  {bytes_alg / best / 1e6:.0f} GB/s  {nq * nd / best / 1e6:.2f} G pair-ops/s", flush=True)
