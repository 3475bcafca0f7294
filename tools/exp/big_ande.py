#!/usr/bin/env python3
"""Lab (not shipped): ande of many queries against 1 Mi rows (2048/40) -- scale check of count -> scan -> fill.
usage: big_ande.py [queries] [unit]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
unit = len(sys.argv) > 2 and sys.argv[2] == "unit"
nd = 1 << 20
q_csr = sdr_amd.synth_csr(2, 0, nq, 2048, 40, unit)
d_csr = sdr_amd.synth_csr(1, 0, nd, 2048, 40, unit)
q = SDRBatch.from_csr(*q_csr, width=2048)
d = SDRBatch.from_csr(*d_csr, width=2048)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = q.ande(d)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"ande {nq} x {nd}: {r.total} elements in {dt * 1e3:.1f} ms", flush=True)
    total = r.total
    del r
# closed form: total elements of EmptyData ande = sum over ids of cntQ[id] * cntD[id]
if not unit:
    hq = np.bincount(q_csr[1], minlength=2048).astype(np.int64)
    hd = np.bincount(d_csr[1], minlength=2048).astype(np.int64)
    assert total == int((hq * hd).sum()), (total, int((hq * hd).sum()))
    print("total matches the closed form")
