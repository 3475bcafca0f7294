#!/usr/bin/env python3
"""Perf lab: kernel-level time of one element op's fill (count and scan excluded) for the config-3 shape.
usage: time_elem.py <op: ore|xore|rme|ande> <queries> [rows] [algo]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch, _lib  # noqa: E402
from sdr_amd._lib import check  # noqa: E402

name, nq = sys.argv[1], int(sys.argv[2])
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
algo = sys.argv[4] if len(sys.argv) > 4 else None
op = {"ande": 0, "ore": 1, "xore": 2, "rme": 3}[name]
lib = _lib.load()
q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, 2048, 40), width=2048)
d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, rows, 2048, 40), width=2048)
flags = q._flags(False, algo)
off = torch.empty(nq * rows + 1, dtype=torch.int32, device="cuda:0")
total = C.c_uint64()
check(lib.sdrb_elem_op_count(q._h, d._h, op, flags & ~(128 | 256 | 512), off.data_ptr(), C.byref(total), None))
ids = torch.empty(max(int(total.value), 1), dtype=torch.int32, device="cuda:0")
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.sdrb_elem_op_fill(q._h, d._h, op, flags, off.data_ptr(), ids.data_ptr(), None, None))
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
nbytes = rows * 164 + nq * 164 + (nq * rows + 1) * 4 + int(total.value) * 4
print(f"{os.environ.get('SDR_AMD_LIB', 'default'):40} {name} q={nq} fill {best*1e3:8.1f} us  elements {int(total.value)}  "
      f"{nbytes / best / 1e6:7.0f} GB/s of in + offsets + out")
